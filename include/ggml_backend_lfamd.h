/*
 * ggml_backend_lfamd.h — the GPU-MODULE boundary: what llamafile/cuda.c dlopen()s as
 * ggml-rocm.so and imports by name (llamafile/cuda.c:726-737), served by the MI355X mat-mul module for the two operators
 * it accelerates (GGML_OP_MUL_MAT, GGML_OP_MUL_MAT_ID; everything else stays with the CPU backend: supports_op).
 *
 * ONE header holds every ggml layout this glue touches.  Those marked IN-TREE are restated from the reference's own
 * patches; those marked RECALLED come from un-vendored upstream llama.cpp @ 8b3befc and must be
 * regenerated from the real headers when the submodule is available.  What can drift silently is NOT trusted: operator
 * and type NUMBERS are resolved at link time through the host's own ggml_op_name / ggml_type_name / ggml_type_size /
 * ggml_blck_size callbacks, and ggml_cuda_link answers false on any mismatch.
 *
 * All 12 entry points use GGML_CALL = __attribute__((ms_abi)): the module is built with -DGGML_MULTIPLATFORM semantics
 * because the host is a Cosmopolitan binary (llamafile/cuda.c:66-67, docs/technical_details.md:80-86).
 */
#ifndef GGML_BACKEND_LFAMD_H_
#define GGML_BACKEND_LFAMD_H_

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#include "llamafile_sgemm.h" /* struct ggml_tensor (RECALLED, kept in that one place) */

#ifdef __cplusplus
extern "C" {
#endif

#define GGML_CALL __attribute__((ms_abi))

typedef uint8_t ggml_guid[16];
typedef ggml_guid *ggml_guid_t;
typedef struct ggml_backend_buffer_type *ggml_backend_buffer_type_t;
typedef struct ggml_backend_buffer *ggml_backend_buffer_t;
typedef struct ggml_backend *ggml_backend_t;
typedef struct ggml_backend_event *ggml_backend_event_t;
typedef void *ggml_backend_graph_plan_t;
typedef void *ggml_backend_buffer_type_context_t;
typedef void *ggml_backend_buffer_context_t;
typedef void *ggml_backend_context_t;

enum ggml_status { GGML_STATUS_ALLOC_FAILED = -2, GGML_STATUS_FAILED = -1, GGML_STATUS_SUCCESS = 0, GGML_STATUS_ABORTED = 1 }; /* RECALLED */
enum ggml_backend_buffer_usage { GGML_BACKEND_BUFFER_USAGE_ANY = 0, GGML_BACKEND_BUFFER_USAGE_WEIGHTS = 1 };                    /* RECALLED */

/* IN-TREE order: ggml-cuda.cu.patch:17087-17094 */
struct ggml_backend_buffer_type_i {
    const char *(*GGML_CALL get_name)(ggml_backend_buffer_type_t buft);
    ggml_backend_buffer_t (*GGML_CALL alloc_buffer)(ggml_backend_buffer_type_t buft, size_t size);
    size_t (*GGML_CALL get_alignment)(ggml_backend_buffer_type_t buft);
    size_t (*GGML_CALL get_max_size)(ggml_backend_buffer_type_t buft);
    size_t (*GGML_CALL get_alloc_size)(ggml_backend_buffer_type_t buft, const struct ggml_tensor *tensor);
    bool (*GGML_CALL is_host)(ggml_backend_buffer_type_t buft);
};
struct ggml_backend_buffer_type { /* RECALLED */
    struct ggml_backend_buffer_type_i iface;
    ggml_backend_buffer_type_context_t context;
};

/* IN-TREE order: ggml-cuda.cu.patch:17017-17027 */
struct ggml_backend_buffer_i {
    const char *(*GGML_CALL get_name)(ggml_backend_buffer_t buffer);
    void (*GGML_CALL free_buffer)(ggml_backend_buffer_t buffer);
    void *(*GGML_CALL get_base)(ggml_backend_buffer_t buffer);
    void (*GGML_CALL init_tensor)(ggml_backend_buffer_t buffer, struct ggml_tensor *tensor);
    void (*GGML_CALL set_tensor)(ggml_backend_buffer_t buffer, struct ggml_tensor *tensor, const void *data, size_t offset, size_t size);
    void (*GGML_CALL get_tensor)(ggml_backend_buffer_t buffer, const struct ggml_tensor *tensor, void *data, size_t offset, size_t size);
    bool (*GGML_CALL cpy_tensor)(ggml_backend_buffer_t buffer, const struct ggml_tensor *src, struct ggml_tensor *dst);
    void (*GGML_CALL clear)(ggml_backend_buffer_t buffer, uint8_t value);
    void (*GGML_CALL reset)(ggml_backend_buffer_t buffer);
};
struct ggml_backend_buffer { /* RECALLED */
    struct ggml_backend_buffer_i iface;
    ggml_backend_buffer_type_t buft;
    ggml_backend_buffer_context_t context;
    size_t size;
    enum ggml_backend_buffer_usage usage;
};

struct ggml_cgraph { /* RECALLED (only size fields and `nodes` are read) */
    int size;
    int n_nodes;
    int n_leafs;
    struct ggml_tensor **nodes;
    struct ggml_tensor **grads;
    struct ggml_tensor **leafs;
    /* visited hash set, eval order: not touched */
};

/* IN-TREE order: ggml-cuda.cu.patch:19479-19500 */
struct ggml_backend_i {
    const char *(*GGML_CALL get_name)(ggml_backend_t backend);
    void (*GGML_CALL free)(ggml_backend_t backend);
    ggml_backend_buffer_type_t (*GGML_CALL get_default_buffer_type)(ggml_backend_t backend);
    void (*GGML_CALL set_tensor_async)(ggml_backend_t backend, struct ggml_tensor *tensor, const void *data, size_t offset, size_t size);
    void (*GGML_CALL get_tensor_async)(ggml_backend_t backend, const struct ggml_tensor *tensor, void *data, size_t offset, size_t size);
    bool (*GGML_CALL cpy_tensor_async)(ggml_backend_t backend_src, ggml_backend_t backend_dst, const struct ggml_tensor *src, struct ggml_tensor *dst);
    void (*GGML_CALL synchronize)(ggml_backend_t backend);
    ggml_backend_graph_plan_t (*GGML_CALL graph_plan_create)(ggml_backend_t backend, const struct ggml_cgraph *cgraph);
    void (*GGML_CALL graph_plan_free)(ggml_backend_t backend, ggml_backend_graph_plan_t plan);
    void (*GGML_CALL graph_plan_update)(ggml_backend_t backend, ggml_backend_graph_plan_t plan, const struct ggml_cgraph *cgraph);
    enum ggml_status (*GGML_CALL graph_plan_compute)(ggml_backend_t backend, ggml_backend_graph_plan_t plan);
    enum ggml_status (*GGML_CALL graph_compute)(ggml_backend_t backend, struct ggml_cgraph *cgraph);
    bool (*GGML_CALL supports_op)(ggml_backend_t backend, const struct ggml_tensor *op);
    bool (*GGML_CALL supports_buft)(ggml_backend_t backend, ggml_backend_buffer_type_t buft);
    bool (*GGML_CALL offload_op)(ggml_backend_t backend, const struct ggml_tensor *op);
    ggml_backend_event_t (*GGML_CALL event_new)(ggml_backend_t backend);
    void (*GGML_CALL event_free)(ggml_backend_event_t event);
    void (*GGML_CALL event_record)(ggml_backend_event_t event);
    void (*GGML_CALL event_wait)(ggml_backend_t backend, ggml_backend_event_t event);
    void (*GGML_CALL event_synchronize)(ggml_backend_event_t event);
};
struct ggml_backend { /* IN-TREE: ggml-cuda.cu.patch:19519-19523 */
    ggml_guid_t guid;
    struct ggml_backend_i iface;
    ggml_backend_context_t context;
};

typedef ggml_backend_t (*GGML_CALL ggml_backend_init_fn)(const char *params, void *user_data);

/* IN-TREE: llama.cpp.patches/patches/ggml-backend-impl.h.patch:20-58 — what the host hands to ggml_cuda_link */
struct ggml_backend_api {
    bool *FLAG_log_disable;
    void (*GGML_CALL exit)(int);
    void (*GGML_CALL free)(void *);
    void *(*GGML_CALL malloc)(size_t);
    char *(*GGML_CALL getenv)(const char *);
    long (*GGML_CALL write)(int, const void *, long);
    void (*GGML_CALL ggml_backend_register)(const char *, ggml_backend_init_fn, ggml_backend_buffer_type_t, void *);
    ggml_backend_buffer_t (*GGML_CALL ggml_backend_buffer_init)(ggml_backend_buffer_type_t, struct ggml_backend_buffer_i, ggml_backend_buffer_context_t, size_t);
    ggml_backend_buffer_t (*GGML_CALL ggml_backend_cpu_buffer_from_ptr)(void *, size_t);
    ggml_backend_buffer_type_t (*GGML_CALL ggml_backend_cpu_buffer_type)(void);
    size_t (*GGML_CALL ggml_backend_buft_get_alloc_size)(ggml_backend_buffer_type_t, struct ggml_tensor *);
    ggml_backend_buffer_t (*GGML_CALL ggml_backend_buft_alloc_buffer)(ggml_backend_buffer_type_t, size_t);
    bool (*GGML_CALL ggml_backend_is_cpu)(ggml_backend_t);
    void (*GGML_CALL ggml_backend_tensor_get)(const struct ggml_tensor *, void *, size_t, size_t);
    void (*GGML_CALL ggml_backend_tensor_set)(struct ggml_tensor *, const void *, size_t, size_t);
    bool (*GGML_CALL ggml_is_quantized)(int);
    size_t (*GGML_CALL ggml_type_size)(int);
    int64_t (*GGML_CALL ggml_blck_size)(int);
    bool (*GGML_CALL ggml_is_transposed)(const struct ggml_tensor *);
    size_t (*GGML_CALL ggml_nbytes)(const struct ggml_tensor *);
    int (*GGML_CALL ggml_get_unary_op)(const struct ggml_tensor *);
    int64_t (*GGML_CALL ggml_nelements)(const struct ggml_tensor *);
    int64_t (*GGML_CALL ggml_nrows)(const struct ggml_tensor *);
    bool (*GGML_CALL ggml_is_permuted)(const struct ggml_tensor *);
    bool (*GGML_CALL ggml_is_contiguous)(const struct ggml_tensor *);
    const char *(*GGML_CALL ggml_op_name)(int);
    const char *(*GGML_CALL ggml_type_name)(int);
    size_t (*GGML_CALL ggml_element_size)(const struct ggml_tensor *);
    size_t (*GGML_CALL ggml_row_size)(int, int64_t);
    void (*GGML_CALL ggml_rope_yarn_corr_dims)(int, int, float, float, float, float[2]);
    const char *(*GGML_CALL ggml_op_desc)(const struct ggml_tensor *);
    bool (*GGML_CALL ggml_backend_buffer_is_host)(ggml_backend_buffer_t);
    bool (*GGML_CALL ggml_guid_matches)(ggml_guid_t, ggml_guid_t);
    bool (*GGML_CALL ggml_is_empty)(const struct ggml_tensor *);
    enum ggml_backend_buffer_usage (*GGML_CALL ggml_backend_buffer_get_usage)(ggml_backend_buffer_t);
    bool (*GGML_CALL ggml_are_same_shape)(const struct ggml_tensor *, const struct ggml_tensor *);
    bool (*GGML_CALL ggml_is_contiguous_1)(const struct ggml_tensor *);
    bool (*GGML_CALL ggml_is_contiguous_2)(const struct ggml_tensor *);
};

/* IN-TREE: llama.cpp.patches/patches/ggml-cuda.h.patch:8-15 */
struct ggml_cuda_device_properties {
    char name[256];
    size_t totalGlobalMem;
    int multiProcessorCount;
    int major;
    int minor;
    char compute[8];
};

/* ---- the 12 symbols llamafile/cuda.c:726-737 imports ----
 * `device` is a LOGICAL device: 0 .. ggml_backend_cuda_get_device_count() - 1, every gfx950 device of the process unless
 * LFAMD_BACKEND_DEVICES (comma list of HIP ordinals, repeats allowed) says otherwise.  ggml_backend_cuda_split_buffer_type is
 * the reference's row split (ggml-cuda.cu.patch:17123-17450) when there is more than one device, the first device's ordinary
 * buffer type otherwise; tensor_split holds GGML_CUDA_MAX_DEVICES = 16 proportions (all zero / NULL: equal shares). */
GGML_CALL bool ggml_cuda_link(const struct ggml_backend_api *backend_api);
GGML_CALL ggml_backend_buffer_type_t ggml_backend_cuda_host_buffer_type(void);
GGML_CALL ggml_backend_buffer_type_t ggml_backend_cuda_buffer_type(int device);
GGML_CALL ggml_backend_t ggml_backend_cuda_init(int device);
GGML_CALL ggml_backend_buffer_type_t ggml_backend_cuda_split_buffer_type(const float *tensor_split);
GGML_CALL int ggml_backend_cuda_reg_devices(void);
GGML_CALL void ggml_backend_cuda_get_device_properties(int device, struct ggml_cuda_device_properties *properties);
GGML_CALL void ggml_backend_cuda_get_device_memory(int device, size_t *free, size_t *total);
GGML_CALL int ggml_backend_cuda_get_device_count(void);
GGML_CALL void ggml_backend_cuda_unregister_host_buffer(void *buffer);
GGML_CALL bool ggml_backend_cuda_register_host_buffer(void *buffer, size_t size);
GGML_CALL void ggml_backend_cuda_get_device_description(int device, char *description, size_t description_size);

#ifdef __cplusplus
}
#endif
#endif
