// Vendor-library comparison point: rocSPARSE CSR SpMV (generic API) on the same matrix, timed
// like the EHYB loop (warm-ups, HIP events around `iters` multiplies, no copies inside).
// Plays the role of the reference's cuSPARSE baseline spmvGeneric (spmv.cu:135-281, disabled
// at its call site solver_test.c:359) with its typing bug fixed (the reference passes
// CUDA_R_32F for fp64 buffers, spmv.cu:186-223).  A tool, NOT part of libehyb.so: the product
// path never links or calls rocSPARSE.
//
// build: hipcc -O3 -shared -fPIC --offload-arch=gfx950 tools/rocsparse_baseline.cpp -lrocsparse -o tools/librocsparse_baseline.so
#include <hip/hip_runtime.h>
#include <rocsparse/rocsparse.h>

#include <cstdint>
#include <cstdio>

#define CHK_HIP(e)                                                                 \
    do {                                                                           \
        hipError_t _e = (e);                                                       \
        if (_e != hipSuccess) {                                                    \
            fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(_e));                \
            return 1;                                                              \
        }                                                                          \
    } while (0)
#define CHK_RS(e)                                                  \
    do {                                                           \
        rocsparse_status _s = (e);                                 \
        if (_s != rocsparse_status_success) {                      \
            fprintf(stderr, "%s: status %d\n", #e, (int)_s);       \
            return 2;                                              \
        }                                                          \
    } while (0)

// alg: 0 default, 2 adaptive, 3 rowsplit, 7 lrb, 8 nnzsplit (rocsparse_spmv_alg).
// y_host receives the product; *ms_per_iter the mean time of one multiply.
extern "C" int rocsparse_csr_spmv_bench(int n, int64_t nnz, const int* rowptr32, const int* col, const double* val,
                                        const double* x_host, double* y_host, int alg, int warmup, int iters,
                                        double* ms_per_iter, double* ms_preprocess)
{
    int* d_rp = nullptr;
    int* d_col = nullptr;
    double *d_val = nullptr, *d_x = nullptr, *d_y = nullptr;
    CHK_HIP(hipMalloc((void**)&d_rp, sizeof(int) * ((size_t)n + 1)));
    CHK_HIP(hipMalloc((void**)&d_col, sizeof(int) * (size_t)nnz));
    CHK_HIP(hipMalloc((void**)&d_val, sizeof(double) * (size_t)nnz));
    CHK_HIP(hipMalloc((void**)&d_x, sizeof(double) * (size_t)n));
    CHK_HIP(hipMalloc((void**)&d_y, sizeof(double) * (size_t)n));
    CHK_HIP(hipMemcpy(d_rp, rowptr32, sizeof(int) * ((size_t)n + 1), hipMemcpyHostToDevice));
    CHK_HIP(hipMemcpy(d_col, col, sizeof(int) * (size_t)nnz, hipMemcpyHostToDevice));
    CHK_HIP(hipMemcpy(d_val, val, sizeof(double) * (size_t)nnz, hipMemcpyHostToDevice));
    CHK_HIP(hipMemcpy(d_x, x_host, sizeof(double) * (size_t)n, hipMemcpyHostToDevice));
    CHK_HIP(hipMemset(d_y, 0, sizeof(double) * (size_t)n));

    rocsparse_handle h;
    CHK_RS(rocsparse_create_handle(&h));
    rocsparse_spmat_descr A;
    rocsparse_dnvec_descr X, Y;
    CHK_RS(rocsparse_create_csr_descr(&A, n, n, nnz, d_rp, d_col, d_val, rocsparse_indextype_i32, rocsparse_indextype_i32,
                                      rocsparse_index_base_zero, rocsparse_datatype_f64_r));
    CHK_RS(rocsparse_create_dnvec_descr(&X, n, d_x, rocsparse_datatype_f64_r));
    CHK_RS(rocsparse_create_dnvec_descr(&Y, n, d_y, rocsparse_datatype_f64_r));
    const double alpha = 1.0, beta = 0.0;
    const rocsparse_spmv_alg a = (rocsparse_spmv_alg)alg;
    size_t bytes = 0;
    void* buf = nullptr;
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wdeprecated-declarations"
    CHK_RS(rocsparse_spmv(h, rocsparse_operation_none, &alpha, A, X, &beta, Y, rocsparse_datatype_f64_r, a,
                          rocsparse_spmv_stage_buffer_size, &bytes, nullptr));
    CHK_HIP(hipMalloc(&buf, bytes ? bytes : 8));
    hipEvent_t e0, e1;
    CHK_HIP(hipEventCreate(&e0));
    CHK_HIP(hipEventCreate(&e1));
    CHK_HIP(hipEventRecord(e0, 0));
    CHK_RS(rocsparse_spmv(h, rocsparse_operation_none, &alpha, A, X, &beta, Y, rocsparse_datatype_f64_r, a,
                          rocsparse_spmv_stage_preprocess, &bytes, buf));
    CHK_HIP(hipEventRecord(e1, 0));
    CHK_HIP(hipEventSynchronize(e1));
    float ms = 0;
    CHK_HIP(hipEventElapsedTime(&ms, e0, e1));
    if (ms_preprocess) *ms_preprocess = ms;
    for (int i = 0; i < warmup; ++i)
        CHK_RS(rocsparse_spmv(h, rocsparse_operation_none, &alpha, A, X, &beta, Y, rocsparse_datatype_f64_r, a,
                              rocsparse_spmv_stage_compute, &bytes, buf));
    CHK_HIP(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; ++i)
        CHK_RS(rocsparse_spmv(h, rocsparse_operation_none, &alpha, A, X, &beta, Y, rocsparse_datatype_f64_r, a,
                              rocsparse_spmv_stage_compute, &bytes, buf));
#pragma clang diagnostic pop
    CHK_HIP(hipEventRecord(e1, 0));
    CHK_HIP(hipEventSynchronize(e1));
    CHK_HIP(hipEventElapsedTime(&ms, e0, e1));
    if (ms_per_iter) *ms_per_iter = ms / iters;
    CHK_HIP(hipMemcpy(y_host, d_y, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost));
    rocsparse_destroy_spmat_descr(A);
    rocsparse_destroy_dnvec_descr(X);
    rocsparse_destroy_dnvec_descr(Y);
    rocsparse_destroy_handle(h);
    hipFree(buf), hipFree(d_rp), hipFree(d_col), hipFree(d_val), hipFree(d_x), hipFree(d_y);
    hipEventDestroy(e0), hipEventDestroy(e1);
    return 0;
}
