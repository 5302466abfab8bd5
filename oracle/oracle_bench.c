/*
 * oracle_bench.c — multi-threaded driver for timing the CPU restatement on the host's cores
 * (bench.py's cpu_baseline leg, kind "port").  TEST INFRASTRUCTURE ONLY.
 *
 * Mirrors how the reference's own harness drives llamafile_sgemm: an OpenMP parallel-for over
 * ith in [0, nth) (llamafile/sgemm_matmul_test.cpp:32-40).
 */
#include "oracle.h"
#include <omp.h>

int ora_sgemm_openmp(long m, long n, long k, const void *A, long lda, const void *B, long ldb,
                     void *C, long ldc, int nth, int Atype, int Btype, int Ctype,
                     const ora_variant *v) {
    int ok = 1;
#pragma omp parallel for num_threads(nth) schedule(static, 1)
    for (int ith = 0; ith < nth; ++ith) {
        int r = ora_llamafile_sgemm(m, n, k, A, lda, B, ldb, C, ldc, ith, nth, Atype, Btype, Ctype, v);
        if (r != 1) {
#pragma omp atomic write
            ok = r;
        }
    }
    return ok;
}

int ora_max_threads(void) {
    return omp_get_max_threads();
}
