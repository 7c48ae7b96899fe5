// Device-resident conjugate gradients on top of the EHYB plan (the iterative caller
// the reference repository was stripped down from).  Reference leftovers this stands in for:
// kernelMyxpy (y = x + gamma*y, kernel.cu:288-296), kernelInitializeAll / kernelInitializeR
// (kernel.cu:20-42), myxpy / initialize_all wrappers (kernel.cu:298-321).
//
// Per iteration: one ehyb_spmv (q = A p) and three memory-bound vector kernels, each a grid-stride
// pass with one wave shuffle + LDS reduction and one fp64 atomic per workgroup:
//   1. pq   = p . q
//   2. x += alpha p ; r -= alpha q ; rz_new = r . z ; rr = r . r   (alpha = rz / pq, read from the device)
//   3. p  = z + beta p                                              (beta  = rz_new / rz)  [kernelMyxpy]
// with z = M^-1 r for the diagonal (Jacobi) preconditioner -- the PRECOND switch of the reference's
// cb_s (spmv.h:7-15) -- recomputed on the fly from 1/diag, or z = r without one.
// The scalars live in one device array; nothing is copied to the host inside the loop except
// the residual norm every `check_every` iterations.
#include <hip/hip_runtime.h>

#include <cmath>

#include "ehyb_internal.h"

using namespace ehyb;

#define CG_TRY(expr)                                                                                        \
    do {                                                                                                    \
        hipError_t _e = (expr);                                                                             \
        if (_e != hipSuccess) {                                                                             \
            ::ehyb::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__);   \
            return EHYB_ERR_HIP;                                                                            \
        }                                                                                                   \
    } while (0)

namespace {

enum { S_RS = 0, S_PQ = 1, S_RS_NEW = 2, S_BB = 3, S_RR = 4 };  // r.z, p.q, next r.z, b.b, r.r

constexpr int kThreads = 256;

__device__ __forceinline__ void block_add(double v, double* __restrict__ target)
{
    __shared__ double part[kThreads / 64];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0.0;
#pragma unroll
        for (int w = 0; w < kThreads / 64; ++w) s += part[w];
        unsafeAtomicAdd(target, s);
    }
}

// r = b - q (q = A x0), z = dinv .* r (or r), p = z, rs = r.z, rr = r.r, bb = b.b
__global__ __launch_bounds__(kThreads) void cg_init_kernel(int n, const double* __restrict__ b,
                                                           const double* __restrict__ q, const double* __restrict__ dinv,
                                                           double* __restrict__ r, double* __restrict__ p,
                                                           double* __restrict__ s)
{
    double rs = 0.0, rr = 0.0, bb = 0.0;
    for (int i = blockIdx.x * kThreads + threadIdx.x; i < n; i += gridDim.x * kThreads) {
        const double bi = b[i], ri = bi - q[i];
        const double zi = dinv ? ri * dinv[i] : ri;
        r[i] = ri;
        p[i] = zi;
        rs = fma(ri, zi, rs);
        rr = fma(ri, ri, rr);
        bb = fma(bi, bi, bb);
    }
    block_add(rs, s + S_RS);
    __syncthreads();
    block_add(bb, s + S_BB);
    __syncthreads();
    block_add(rr, s + S_RR + 1);  // the slot the host reads; S_RR itself stays zero for the first iteration
}

__global__ __launch_bounds__(kThreads) void cg_dot_kernel(int n, const double* __restrict__ p,
                                                          const double* __restrict__ q, double* __restrict__ s)
{
    double acc = 0.0;
    for (int i = blockIdx.x * kThreads + threadIdx.x; i < n; i += gridDim.x * kThreads) acc = fma(p[i], q[i], acc);
    block_add(acc, s + S_PQ);
}

__global__ __launch_bounds__(kThreads) void cg_update_kernel(int n, const double* __restrict__ p,
                                                             const double* __restrict__ q, const double* __restrict__ dinv,
                                                             double* __restrict__ x, double* __restrict__ r,
                                                             double* __restrict__ s)
{
    const double alpha = s[S_RS] / s[S_PQ];
    double rz = 0.0, rr = 0.0;
    for (int i = blockIdx.x * kThreads + threadIdx.x; i < n; i += gridDim.x * kThreads) {
        x[i] = fma(alpha, p[i], x[i]);
        const double ri = fma(-alpha, q[i], r[i]);
        r[i] = ri;
        rz = fma(ri, dinv ? ri * dinv[i] : ri, rz);
        rr = fma(ri, ri, rr);
    }
    block_add(rz, s + S_RS_NEW);
    __syncthreads();
    block_add(rr, s + S_RR);
}

// p = z + beta p  (the reference's kernelMyxpy with gamma = beta), then roll the scalars
__global__ __launch_bounds__(kThreads) void cg_direction_kernel(int n, const double* __restrict__ r,
                                                                const double* __restrict__ dinv, double* __restrict__ p,
                                                                const double* __restrict__ s)
{
    const double beta = s[S_RS_NEW] / s[S_RS];
    for (int i = blockIdx.x * kThreads + threadIdx.x; i < n; i += gridDim.x * kThreads)
        p[i] = fma(beta, p[i], dinv ? r[i] * dinv[i] : r[i]);
}

__global__ void cg_roll_kernel(double* __restrict__ s)
{
    s[S_RS] = s[S_RS_NEW];
    s[S_RS_NEW] = 0.0;
    s[S_PQ] = 0.0;
    s[S_RR + 1] = s[S_RR];  // the residual norm the host reads
    s[S_RR] = 0.0;
}

}  // namespace

extern "C" int ehyb_cg(ehyb_plan* P, const double* b, double* x, int max_iter, double rtol, int check_every,
                       void* stream, int* iters_done, double* rel_residual)
{
    return ehyb_pcg(P, nullptr, b, x, max_iter, rtol, check_every, stream, iters_done, rel_residual);
}

extern "C" int ehyb_pcg(ehyb_plan* P, const double* dinv, const double* b, double* x, int max_iter, double rtol,
                        int check_every, void* stream, int* iters_done, double* rel_residual)
{
    clear_error();
    if (!P || !b || !x || max_iter < 0 || !(rtol >= 0)) EHYB_FAIL(EHYB_ERR_ARG, "ehyb_cg: bad arguments");
    if (!P->uploaded) EHYB_FAIL(EHYB_ERR_STATE, "ehyb_cg: plan not uploaded (no CPU fallback exists)");
    if (P->host.row_begin != 0 || P->host.row_end != P->host.n_cols)
        EHYB_FAIL(EHYB_ERR_ARG, "ehyb_cg: needs a plan over all rows");
    const int n = P->host.n_cols;
    hipStream_t st = (hipStream_t)stream;
    if (check_every <= 0) check_every = 10;
    double *r = nullptr, *p = nullptr, *q = nullptr, *s = nullptr;
    auto cleanup = [&]() {
        if (r) (void)hipFree(r);
        if (p) (void)hipFree(p);
        if (q) (void)hipFree(q);
        if (s) (void)hipFree(s);
    };
    CG_TRY(hipMalloc((void**)&r, (size_t)n * 8));
    CG_TRY(hipMalloc((void**)&p, (size_t)n * 8));
    CG_TRY(hipMalloc((void**)&q, (size_t)n * 8));
    CG_TRY(hipMalloc((void**)&s, 8 * sizeof(double)));
    CG_TRY(hipMemsetAsync(s, 0, 8 * sizeof(double), st));
    const int grid = std::max(1, std::min((n + kThreads - 1) / kThreads, 2048));

    int rc = ehyb_spmv(P, x, q, stream);  // q = A x0
    if (rc != EHYB_OK) {
        cleanup();
        return rc;
    }
    hipLaunchKernelGGL(cg_init_kernel, dim3(grid), dim3(kThreads), 0, st, n, b, q, dinv, r, p, s);
    double h[6] = {0, 0, 0, 0, 0, 0};
    CG_TRY(hipMemcpyAsync(h, s, sizeof h, hipMemcpyDeviceToHost, st));
    CG_TRY(hipStreamSynchronize(st));
    const double bb = h[S_BB] > 0 ? h[S_BB] : 1.0;
    double rs = h[S_RR + 1];  // ||r||^2 (the preconditioned product r.z drives the recurrences, not the stop test)
    int it = 0;
    while (it < max_iter && std::sqrt(rs / bb) > rtol) {
        const int burst = std::min(check_every, max_iter - it);
        for (int k = 0; k < burst; ++k) {
            if ((rc = ehyb_spmv(P, p, q, stream)) != EHYB_OK) {  // q = A p: x of the multiply changes every time
                cleanup();
                return rc;
            }
            hipLaunchKernelGGL(cg_dot_kernel, dim3(grid), dim3(kThreads), 0, st, n, p, q, s);
            hipLaunchKernelGGL(cg_update_kernel, dim3(grid), dim3(kThreads), 0, st, n, p, q, dinv, x, r, s);
            hipLaunchKernelGGL(cg_direction_kernel, dim3(grid), dim3(kThreads), 0, st, n, r, dinv, p, s);
            hipLaunchKernelGGL(cg_roll_kernel, dim3(1), dim3(1), 0, st, s);
        }
        it += burst;
        CG_TRY(hipGetLastError());
        CG_TRY(hipMemcpyAsync(h, s, sizeof h, hipMemcpyDeviceToHost, st));
        CG_TRY(hipStreamSynchronize(st));
        rs = h[S_RR + 1];
        if (!(rs == rs) || !(h[S_RS] == h[S_RS])) {
            rs = NAN;
            break;  // NaN: breakdown (matrix or preconditioner not positive definite)
        }
    }
    cleanup();
    if (iters_done) *iters_done = it;
    if (rel_residual) *rel_residual = std::sqrt(rs / bb);
    if (!(rs == rs)) EHYB_FAIL(EHYB_ERR_ARG, "ehyb_cg: breakdown (is the matrix symmetric positive definite?)");
    return EHYB_OK;
}
