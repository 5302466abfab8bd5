// blaslt.hip — batches on PLAIN 16-bit float weight matrices through the vendor's GEMM (hipBLASLt, dlopen()ed like librccl in
// comm.hip): F16 / BF16 weight tensors (tinyBLAS float types, tinyblas_cpu.h:419-613) and the resident f16 image of Q8_0
// weights (one f16 rounding of d * q per weight, one of d8 * code per activation: the scaled-operand numerics of the K-quant
// batches, <= 1e-3).  No dequantisation, no fused prologue: this is a library GEMM, what the vendor library is for; the
// hand-written kernels are the ones that read quantised bytes.  C (m x n, column-major, ld = ldc) = W (m x k, rows) x X^T.
//
//   lfamd_blaslt_ok()                       the library is loadable and a handle exists (else callers use their own bodies)
//   lfamd_blaslt_gemm(dtype, W, ldw, X, ldx, m, n, k, C, ldc, ws, ws_bytes, stream)
//   lfamd_launch_rows_to_f16 / _q80_rows_to_f16 / _q80_image     the row-major 16-bit operands
#include "lfamd_device.h"

#include <dlfcn.h>
#include <stdlib.h>
#include <hipblaslt/hipblaslt.h>

#include <map>
#include <set>
#include <mutex>
#include <tuple>

namespace {

struct lt_api {
    void *so = nullptr;
    hipblasLtHandle_t handle = nullptr;      // of the device current at load time
    hipblasLtHandle_t per_device[32] = {};   // one handle per device (made with that device current, on first use there)
    int first_device = 0;
    decltype(&hipblasLtCreate) Create = nullptr;
    decltype(&hipblasLtMatmulDescCreate) DescCreate = nullptr;
    decltype(&hipblasLtMatmulDescSetAttribute) DescSet = nullptr;
    decltype(&hipblasLtMatmulDescDestroy) DescDestroy = nullptr;
    decltype(&hipblasLtMatrixLayoutCreate) LayoutCreate = nullptr;
    decltype(&hipblasLtMatrixLayoutDestroy) LayoutDestroy = nullptr;
    decltype(&hipblasLtMatmulPreferenceCreate) PrefCreate = nullptr;
    decltype(&hipblasLtMatmulPreferenceSetAttribute) PrefSet = nullptr;
    decltype(&hipblasLtMatmulPreferenceDestroy) PrefDestroy = nullptr;
    decltype(&hipblasLtMatmulAlgoGetHeuristic) Heuristic = nullptr;
    decltype(&hipblasLtMatmul) Matmul = nullptr;
    bool ok = false;
};

struct plan {
    hipblasLtMatmulDesc_t desc = nullptr;
    hipblasLtMatrixLayout_t a = nullptr, b = nullptr, c = nullptr;
    hipblasLtMatmulAlgo_t algo;
    size_t ws = 0;
    bool ok = false;
};

lt_api g_lt;
std::once_flag g_once;
std::mutex g_mu;
std::map<std::tuple<int, long, long, long, long, long, long>, plan> g_plans;

template <typename F>
bool sym(void *so, const char *name, F &f) {
    f = (F)dlsym(so, name);
    return f != nullptr;
}

void load() {
    // opt-in (round 4): the module's own kernels are the default for every type; LFAMD_USE_BLASLT=1 selects the vendor GEMM for
    // plain 16-bit float weights and Q8_0 batches (a yardstick, and a second resident image for Q8_0: lfamd_packed_size)
    const char *use = getenv("LFAMD_USE_BLASLT");
    if (!use || atoi(use) == 0 || getenv("LFAMD_NO_BLASLT"))
        return;
    int devices = 0; // (hipblasLtCreate ends the process when there is no device)
    if (hipGetDeviceCount(&devices) != hipSuccess || devices <= 0) {
        (void)hipGetLastError();
        return;
    }
    const char *names[] = {"libhipblaslt.so.1", "libhipblaslt.so", "/opt/rocm/lib/libhipblaslt.so.1"};
    for (const char *n : names)
        if ((g_lt.so = dlopen(n, RTLD_NOW | RTLD_LOCAL)))
            break;
    if (!g_lt.so)
        return;
    lt_api &a = g_lt;
    if (!(sym(a.so, "hipblasLtCreate", a.Create) && sym(a.so, "hipblasLtMatmulDescCreate", a.DescCreate) &&
          sym(a.so, "hipblasLtMatmulDescSetAttribute", a.DescSet) && sym(a.so, "hipblasLtMatmulDescDestroy", a.DescDestroy) &&
          sym(a.so, "hipblasLtMatrixLayoutCreate", a.LayoutCreate) && sym(a.so, "hipblasLtMatrixLayoutDestroy", a.LayoutDestroy) &&
          sym(a.so, "hipblasLtMatmulPreferenceCreate", a.PrefCreate) && sym(a.so, "hipblasLtMatmulPreferenceSetAttribute", a.PrefSet) &&
          sym(a.so, "hipblasLtMatmulPreferenceDestroy", a.PrefDestroy) && sym(a.so, "hipblasLtMatmulAlgoGetHeuristic", a.Heuristic) &&
          sym(a.so, "hipblasLtMatmul", a.Matmul)))
        return;
    if (hipGetDevice(&a.first_device) != hipSuccess || a.first_device < 0 || a.first_device >= 32)
        return;
    if (a.Create(&a.handle) != HIPBLAS_STATUS_SUCCESS)
        return;
    a.per_device[a.first_device] = a.handle;
    a.ok = true;
}

// f32 rows -> f16 / bf16 rows (row-major, k halves per row)
template <bool BF>
__global__ void rows_to_16_kernel(const uint8_t *__restrict__ X, size_t x_row_bytes, long n, long k, uint16_t *__restrict__ out) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; // one thread per four values
    const long per_row = k / 4;
    if (idx >= n * per_row)
        return;
    const long r = idx / per_row, c = (idx - r * per_row) * 4;
    const float4 f = *(const float4 *)((const float *)(X + (size_t)r * x_row_bytes) + c);
    const float v[4] = {f.x, f.y, f.z, f.w};
    uint16_t o[4];
#pragma unroll
    for (int e = 0; e < 4; e++) {
        if constexpr (BF) { // round to nearest even like the reference's GGML_FP32_TO_BF16 (NaN kept quiet)
            uint32_t u = __builtin_bit_cast(uint32_t, v[e]);
            o[e] = (u & 0x7fffffffu) > 0x7f800000u ? (uint16_t)((u >> 16) | 64) : (uint16_t)((u + (0x7fffu + ((u >> 16) & 1))) >> 16);
        } else {
            o[e] = __builtin_bit_cast(uint16_t, (_Float16)v[e]);
        }
    }
    *(uint2 *)(out + r * k + c) = make_uint2(o[0] | ((uint32_t)o[1] << 16), o[2] | ((uint32_t)o[3] << 16));
}

// Q8_0 activations as f16(d * code), row-major.  F32IN: quantize_row_q8_0 first (d = amax / 127 kept as f16, code = roundf(x / d)).
// One thread per 32-block quarter (8 values); the block maximum over the four threads of a block by DPP.
template <bool F32IN>
__global__ void q80_rows_to_f16_kernel(const uint8_t *__restrict__ X, size_t x_row_bytes, long n, long k, _Float16 *__restrict__ out) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long per_row = k / 8;
    const bool live = idx < n * per_row;
    const long r = live ? idx / per_row : 0, c8 = live ? idx - r * per_row : 0; // eight values c8 * 8 .. + 7 of row r
    float v[8];
    float d;
    if constexpr (F32IN) {
        const float4 *p = (const float4 *)((const float *)(X + (size_t)r * x_row_bytes) + c8 * 8);
        const float4 f0 = p[0], f1 = p[1];
        v[0] = f0.x, v[1] = f0.y, v[2] = f0.z, v[3] = f0.w, v[4] = f1.x, v[5] = f1.y, v[6] = f1.z, v[7] = f1.w;
        float am = 0.0f;
#pragma unroll
        for (int e = 0; e < 8; e++)
            am = fmaxf(am, fabsf(v[e]));
        am = fmaxf(am, dpp_f32<DPP_XOR1>(am)); // the block's four threads are one quad
        am = fmaxf(am, dpp_f32<DPP_XOR2>(am));
        const float dd = am / 127.0f;
        const float id = dd != 0.0f ? 1.0f / dd : 0.0f;
        d = (float)(_Float16)dd; // the scale as the block stores it
#pragma unroll
        for (int e = 0; e < 8; e++)
            v[e] = roundf(v[e] * id);
    } else {
        const lfamd_block_q8_0 *b = (const lfamd_block_q8_0 *)(X + (size_t)r * x_row_bytes) + (c8 >> 2);
        d = h2f(b->d);
#pragma unroll
        for (int e = 0; e < 8; e++)
            v[e] = (float)b->qs[(c8 & 3) * 8 + e];
    }
    if (!live)
        return;
    half8_t o;
#pragma unroll
    for (int e = 0; e < 8; e++)
        o[e] = (_Float16)(d * v[e]);
    *(half8_t *)(out + r * k + c8 * 8) = o;
}

// Q8_0 weights (raw GGUF rows) -> f16(d * q), row-major [rows][cols]
__global__ void q80_image_kernel(const uint8_t *__restrict__ raw, size_t raw_row_bytes, long rows, long cols, _Float16 *__restrict__ out) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; // one thread per 8 weights
    const long per_row = cols / 8;
    if (idx >= rows * per_row)
        return;
    const long r = idx / per_row, c8 = idx - r * per_row;
    const lfamd_block_q8_0 *b = (const lfamd_block_q8_0 *)(raw + (size_t)r * raw_row_bytes) + (c8 >> 2);
    const float d = h2f(b->d);
    half8_t o;
#pragma unroll
    for (int e = 0; e < 8; e++)
        o[e] = (_Float16)(d * (float)b->qs[(c8 & 3) * 8 + e]);
    *(half8_t *)(out + r * cols + c8 * 8) = o;
}

} // namespace

extern "C" {

bool lfamd_blaslt_ok() {
    std::call_once(g_once, load);
    return g_lt.ok;
}

#define LFAMD_BLASLT_WS (32u << 20)
size_t lfamd_blaslt_workspace() {
    return LFAMD_BLASLT_WS;
}

// One product under the lock: hipErrorNotSupported = the library has nothing for this shape on this device.
static void destroy_plan(lt_api &a, plan &q) {
    if (q.desc)
        a.DescDestroy(q.desc);
    for (hipblasLtMatrixLayout_t l : {q.a, q.b, q.c})
        if (l)
            a.LayoutDestroy(l);
    q = plan();
}

static hipError_t blaslt_gemm_one(int dtype, const void *W, long ldw, const void *X, long ldx, long m, long n, long k, float *C, long ldc,
                                  void *ws, size_t ws_bytes, hipStream_t s) {
    lt_api &a = g_lt;
    std::lock_guard<std::mutex> lk(g_mu);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 32)
        return hipErrorInvalidDevice;
    if (!a.per_device[dev] && a.Create(&a.per_device[dev]) != HIPBLAS_STATUS_SUCCESS) {
        a.per_device[dev] = nullptr;
        return hipErrorNotSupported;
    }
    const hipblasLtHandle_t handle = a.per_device[dev];
    // (the device is part of the key: an algorithm is chosen by, and replayed on, one device's handle)
    const auto key = std::make_tuple(dtype * 32 + dev, m, n, k, ldw, ldx, ldc * 2 + (ws_bytes ? 1 : 0));
    static std::set<std::tuple<int, long, long, long, long, long, long>> refused; // shapes the heuristic had nothing for (no objects kept)
    if (refused.count(key))
        return hipErrorNotSupported;
    if (g_plans.size() + refused.size() >= 1024 && !g_plans.count(key)) { // (a KV cache as the A operand grows by a row per token: bound the cache)
        for (auto &kv : g_plans)
            destroy_plan(a, kv.second);
        g_plans.clear();
        refused.clear();
    }
    plan &p = g_plans[key];
    if (!p.desc) { // first use of this shape: descriptors + the heuristic's first choice
        const hipDataType dt = dtype == LFAMD_TYPE_BF16 ? HIP_R_16BF : HIP_R_16F;
        const int32_t opT = HIPBLAS_OP_T, opN = HIPBLAS_OP_N;
        hipblasLtMatmulPreference_t pref = nullptr;
        hipblasLtMatmulHeuristicResult_t res[8];
        int found = 0;
        const uint64_t wsb = ws_bytes;
        // column-major view: W rows = a k x m matrix (ld = ldw), transposed; X rows = a k x n matrix (ld = ldx); C m x n (ld = ldc)
        if (a.DescCreate(&p.desc, HIPBLAS_COMPUTE_32F, HIP_R_32F) == HIPBLAS_STATUS_SUCCESS &&
            a.DescSet(p.desc, HIPBLASLT_MATMUL_DESC_TRANSA, &opT, sizeof opT) == HIPBLAS_STATUS_SUCCESS &&
            a.DescSet(p.desc, HIPBLASLT_MATMUL_DESC_TRANSB, &opN, sizeof opN) == HIPBLAS_STATUS_SUCCESS &&
            a.LayoutCreate(&p.a, dt, (uint64_t)k, (uint64_t)m, ldw) == HIPBLAS_STATUS_SUCCESS &&
            a.LayoutCreate(&p.b, dt, (uint64_t)k, (uint64_t)n, ldx) == HIPBLAS_STATUS_SUCCESS &&
            a.LayoutCreate(&p.c, HIP_R_32F, (uint64_t)m, (uint64_t)n, ldc) == HIPBLAS_STATUS_SUCCESS &&
            a.PrefCreate(&pref) == HIPBLAS_STATUS_SUCCESS &&
            a.PrefSet(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &wsb, sizeof wsb) == HIPBLAS_STATUS_SUCCESS &&
            a.Heuristic(handle, p.desc, p.a, p.b, p.c, p.c, pref, 8, res, &found) == HIPBLAS_STATUS_SUCCESS) {
            for (int r = 0; r < found && !p.ok; r++) // in order of increasing estimated time: the first that fits the workspace
                if (res[r].state == HIPBLAS_STATUS_SUCCESS && res[r].workspaceSize <= ws_bytes) {
                    p.algo = res[r].algo;
                    p.ws = res[r].workspaceSize;
                    p.ok = true;
                }
        }
        if (pref)
            a.PrefDestroy(pref);
        if (!p.ok) { // nothing usable: no half-built plan stays behind (its layouts would leak and the shape would stay refused for good)
            destroy_plan(a, p);
            g_plans.erase(key);
            refused.insert(key);
            return hipErrorNotSupported;
        }
    }
    const float one = 1.0f, zero = 0.0f;
    const hipblasStatus_t st = a.Matmul(handle, p.desc, &one, W, p.a, X, p.b, &zero, C, p.c, C, p.c, &p.algo, ws, p.ws, s);
    return st == HIPBLAS_STATUS_SUCCESS ? hipSuccess : hipErrorUnknown;
}

// dtype: LFAMD_TYPE_F16 or LFAMD_TYPE_BF16 (both operands); C f32.  ws may be null (then only algorithms without workspace).
hipError_t lfamd_blaslt_gemm(int dtype, const void *W, long ldw, const void *X, long ldx, long m, long n, long k, float *C, long ldc,
                             void *ws, size_t ws_bytes, hipStream_t s) {
    if (!lfamd_blaslt_ok())
        return hipErrorNotSupported;
    if (ws_bytes > LFAMD_BLASLT_WS)
        ws_bytes = LFAMD_BLASLT_WS;
    if (!ws)
        ws_bytes = 0;
    hipError_t e = blaslt_gemm_one(dtype, W, ldw, X, ldx, m, n, k, C, ldc, ws, ws_bytes, s);
    if (e != hipErrorNotSupported || m <= 8192)
        return e;
    // a very tall matrix (output.weight) the heuristic has nothing for: the same product in row blocks (the lock is taken per block)
    e = hipSuccess;
    for (long r0 = 0; r0 < m && e == hipSuccess; r0 += 8192) {
        const long mr = m - r0 < 8192 ? m - r0 : 8192;
        e = blaslt_gemm_one(dtype, (const uint8_t *)W + (size_t)r0 * ldw * 2, ldw, X, ldx, mr, n, k, C + r0, ldc, ws, ws_bytes, s);
    }
    return e;
}

hipError_t lfamd_launch_rows_to_16(int dtype, const void *X, size_t x_row_bytes, long n, long k, void *out, hipStream_t s) {
    const long threads = n * (k / 4);
    if (threads == 0)
        return hipSuccess;
    const unsigned grid = (unsigned)((threads + 255) / 256);
    if (dtype == LFAMD_TYPE_BF16)
        rows_to_16_kernel<true><<<grid, 256, 0, s>>>((const uint8_t *)X, x_row_bytes, n, k, (uint16_t *)out);
    else
        rows_to_16_kernel<false><<<grid, 256, 0, s>>>((const uint8_t *)X, x_row_bytes, n, k, (uint16_t *)out);
    return hipGetLastError();
}

hipError_t lfamd_launch_q80_rows_to_f16(int Btype, const void *X, size_t x_row_bytes, long n, long k, void *out, hipStream_t s) {
    const long threads = n * (k / 8);
    if (threads == 0)
        return hipSuccess;
    const unsigned grid = (unsigned)((threads + 255) / 256);
    if (Btype == LFAMD_TYPE_F32)
        q80_rows_to_f16_kernel<true><<<grid, 256, 0, s>>>((const uint8_t *)X, x_row_bytes, n, k, (_Float16 *)out);
    else
        q80_rows_to_f16_kernel<false><<<grid, 256, 0, s>>>((const uint8_t *)X, x_row_bytes, n, k, (_Float16 *)out);
    return hipGetLastError();
}

hipError_t lfamd_launch_q80_image(const void *raw, size_t raw_row_bytes, long rows, long cols, void *out, hipStream_t s) {
    const long threads = rows * (cols / 8);
    if (threads == 0)
        return hipSuccess;
    q80_image_kernel<<<(unsigned)((threads + 255) / 256), 256, 0, s>>>((const uint8_t *)raw, raw_row_bytes, rows, cols, (_Float16 *)out);
    return hipGetLastError();
}
}
