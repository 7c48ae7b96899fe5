// Device-resident conjugate gradients on top of the EHYB plan (the iterative caller
// the reference repository was stripped down from).  Reference leftovers this stands in for:
// kernelMyxpy (y = x + gamma*y, kernel.cu:288-296), kernelInitializeAll / kernelInitializeR
// (kernel.cu:20-42), myxpy / initialize_all wrappers (kernel.cu:298-321).
//
// Per iteration: one ehyb_spmv (q = A p) and three memory-bound vector kernels, each a grid-stride
// pass over the vectors:
//   1. pq   = p . q
//   2. x += alpha p ; r -= alpha q ; rz_new = r . z ; rr = r . r   (alpha = rz / pq)
//   3. p  = z + beta p                                              (beta  = rz_new / rz)  [kernelMyxpy]
// with z = M^-1 r for the diagonal (Jacobi) preconditioner -- the PRECOND switch of the reference's
// cb_s (spmv.h:7-15) -- recomputed on the fly from 1/diag, or z = r without one.
//
// Dot products never leave the device and use no atomics: a kernel writes one partial sum per
// workgroup, and the kernel that needs the scalar adds the (at most 1024) partials up again in a
// fixed order -- every workgroup for itself, 8 KiB out of L2.  (A first version added the partials
// with one fp64 atomic per workgroup: 2048 same-address atomics cost 28 us per dot product,
// profiles/r01_j_cg.txt.)  Nothing has to be zeroed or rolled between iterations; r.z alternates
// between two partial arrays, so an even and an odd iteration differ in one kernel argument and
// are captured together into one hipGraph, replayed once per two iterations.  The sums have a
// fixed order, so a solve is reproducible run to run wherever the multiply is (plain storage).
// The host reads the partials every `check_every` iterations for the stopping test.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdlib>
#include <vector>

#include "ehyb_internal.h"

using namespace ehyb;

#define HIP_TRY(expr)                                                                                       \
    do {                                                                                                    \
        hipError_t _e = (expr);                                                                             \
        if (_e != hipSuccess) {                                                                             \
            ::ehyb::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__);   \
            return EHYB_ERR_HIP;                                                                            \
        }                                                                                                   \
    } while (0)

namespace {

#ifndef EHYB_CG_THREADS
#define EHYB_CG_THREADS 256
#endif
constexpr int kThreads = EHYB_CG_THREADS;
constexpr int kMaxGrid = 1024;  // partial sums per dot product

// partial arrays, kMaxGrid doubles each
// (r.r sits between the two r.z slots, so that the pair an iteration writes -- its new r.z and r.r -- is one
// contiguous range for a multi-GPU caller's all-reduce: slot of r.z number c = A_RZ0 + 2 c)
enum { A_BB = 0, A_PQ = 1, A_RZ0 = 2, A_RR = 3, A_RZ1 = 4, A_COUNT = 5 };

// sum over the workgroup, returned to every thread; fixed order
__device__ __forceinline__ double block_sum(double v)
{
    __shared__ double part[kThreads / 64];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    __syncthreads();  // a previous call's readers are done with part[]
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < kThreads / 64; ++w) s += part[w];
    return s;
}

__device__ __forceinline__ double sum_partials(const double* __restrict__ part)
{
    double v = 0.0;
    for (int i = threadIdx.x; i < (int)gridDim.x; i += kThreads) v += part[i];
    return block_sum(v);
}

__device__ __forceinline__ void put_partial(double v, double* __restrict__ part)
{
    const double s = block_sum(v);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}

// r = b - q (q = A x0), z = dinv .* r (or r), p = z; partials of r.z, r.r, b.b
__global__ __launch_bounds__(kThreads) void cg_init_kernel(int n, const double* __restrict__ b,
                                                           const double* __restrict__ q, const double* __restrict__ dinv,
                                                           double* __restrict__ r, double* __restrict__ p,
                                                           double* __restrict__ s)
{
    double rz = 0.0, rr = 0.0, bb = 0.0;
    for (int i = blockIdx.x * kThreads + threadIdx.x; i < n; i += gridDim.x * kThreads) {
        const double bi = b[i], ri = bi - q[i];
        const double zi = dinv ? ri * dinv[i] : ri;
        r[i] = ri;
        p[i] = zi;
        rz = fma(ri, zi, rz);
        rr = fma(ri, ri, rr);
        bb = fma(bi, bi, bb);
    }
    put_partial(rz, s + A_RZ0 * kMaxGrid);
    put_partial(rr, s + A_RR * kMaxGrid);
    put_partial(bb, s + A_BB * kMaxGrid);
}

__global__ __launch_bounds__(kThreads) void cg_dot_kernel(int n, const double* __restrict__ p,
                                                          const double* __restrict__ q, double* __restrict__ s)
{
    double acc = 0.0;
    for (int i = blockIdx.x * kThreads + threadIdx.x; i < n; i += gridDim.x * kThreads) acc = fma(p[i], q[i], acc);
    put_partial(acc, s + A_PQ * kMaxGrid);
}

// cur: which of the two r.z arrays holds this iteration's r.z; the new one goes to the other
__global__ __launch_bounds__(kThreads) void cg_update_kernel(int n, const double* __restrict__ p,
                                                             const double* __restrict__ q, const double* __restrict__ dinv,
                                                             double* __restrict__ x, double* __restrict__ r,
                                                             double* __restrict__ s, int cur)
{
    const double alpha = sum_partials(s + (A_RZ0 + 2 * cur) * kMaxGrid) / sum_partials(s + A_PQ * kMaxGrid);
    double rz = 0.0, rr = 0.0;
    // four grid strides per trip: sixteen (twenty with a preconditioner) loads in flight per thread instead of four -- a thread
    // sees only seven elements of the bench matrix's vectors, and one load round trip per element was most of this kernel's time
    const int stride = (int)gridDim.x * kThreads;
    int i = blockIdx.x * kThreads + threadIdx.x;
    for (; i + 3 * stride < n; i += 4 * stride) {
        double pv[4], qv[4], xv[4], rv[4], dv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            pv[u] = p[i + u * stride];
            qv[u] = __builtin_nontemporal_load(&q[i + u * stride]);   // q is dead after this kernel, x is not read again before the next
            xv[u] = __builtin_nontemporal_load(&x[i + u * stride]);   // update: streamed past the caches, which hold the matrix's tail
            rv[u] = r[i + u * stride];
            dv[u] = dinv ? dinv[i + u * stride] : 1.0;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            __builtin_nontemporal_store(fma(alpha, pv[u], xv[u]), &x[i + u * stride]);
            const double ri = fma(-alpha, qv[u], rv[u]);
            r[i + u * stride] = ri;
            rz = fma(ri, dinv ? ri * dv[u] : ri, rz);
            rr = fma(ri, ri, rr);
        }
    }
    for (; i < n; i += stride) {
        x[i] = fma(alpha, p[i], x[i]);
        const double ri = fma(-alpha, q[i], r[i]);
        r[i] = ri;
        rz = fma(ri, dinv ? ri * dinv[i] : ri, rz);
        rr = fma(ri, ri, rr);
    }
    put_partial(rz, s + (A_RZ0 + 2 * (cur ^ 1)) * kMaxGrid);
    put_partial(rr, s + A_RR * kMaxGrid);
}

// p = z + beta p  (the reference's kernelMyxpy with gamma = beta)
__global__ __launch_bounds__(kThreads) void cg_direction_kernel(int n, const double* __restrict__ r,
                                                                const double* __restrict__ dinv, double* __restrict__ p,
                                                                const double* __restrict__ s, int cur)
{
    const double beta = sum_partials(s + (A_RZ0 + 2 * (cur ^ 1)) * kMaxGrid) / sum_partials(s + (A_RZ0 + 2 * cur) * kMaxGrid);
    const int stride = (int)gridDim.x * kThreads;
    int i = blockIdx.x * kThreads + threadIdx.x;
    for (; i + 3 * stride < n; i += 4 * stride) {  // (as in the update kernel: the loads of four strides in flight together)
        double pv[4], rv[4], dv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            pv[u] = p[i + u * stride];
            rv[u] = r[i + u * stride];
            dv[u] = dinv ? dinv[i + u * stride] : 1.0;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) p[i + u * stride] = fma(beta, pv[u], dinv ? rv[u] * dv[u] : rv[u]);
    }
    for (; i < n; i += stride) p[i] = fma(beta, p[i], dinv ? r[i] * dinv[i] : r[i]);
}

// everything a solve owns; released on every way out
struct Workspace {
    double *r = nullptr, *p = nullptr, *q = nullptr, *s = nullptr;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    hipStream_t own = nullptr;
    ~Workspace()
    {
        if (exec) (void)hipGraphExecDestroy(exec);
        if (graph) (void)hipGraphDestroy(graph);
        if (own) (void)hipStreamDestroy(own);
        if (r) (void)hipFree(r);
        if (p) (void)hipFree(p);
        if (q) (void)hipFree(q);
        if (s) (void)hipFree(s);
    }
};

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
    check_every += check_every & 1;  // iterations are issued in even/odd pairs
    Workspace W;
    if (!st) {  // the legacy default stream cannot be captured: solve on a private (blocking) stream instead
        HIP_TRY(hipStreamCreate(&W.own));
        st = W.own;
        stream = (void*)W.own;
    }
    HIP_TRY(hipMalloc((void**)&W.r, (size_t)n * 8));
    HIP_TRY(hipMalloc((void**)&W.p, (size_t)n * 8));
    HIP_TRY(hipMalloc((void**)&W.q, (size_t)n * 8));
    HIP_TRY(hipMalloc((void**)&W.s, (size_t)A_COUNT * kMaxGrid * sizeof(double)));
    double *r = W.r, *p = W.p, *q = W.q, *s = W.s;
    // two workgroups per CU: 115.7 us per iteration on the audikw_1-like matrix against 118-120 with
    // 1024 workgroups (twice the partials to re-add) and 126 with 256
    const int grid = std::max(1, std::min((n + kThreads - 1) / kThreads, kMaxGrid / 2));

    int rc = ehyb_spmv(P, x, q, stream);  // q = A x0
    if (rc != EHYB_OK) return rc;
    hipLaunchKernelGGL(cg_init_kernel, dim3(grid), dim3(kThreads), 0, st, n, b, q, dinv, r, p, s);
    std::vector<double> h((size_t)A_COUNT * kMaxGrid);
    auto read_scalar = [&](int which) {  // fixed order, like the device
        double t = 0.0;
        for (int i = 0; i < grid; ++i) t += h[(size_t)which * kMaxGrid + i];
        return t;
    };
    HIP_TRY(hipMemcpyAsync(h.data(), s, h.size() * sizeof(double), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    const double bb0 = read_scalar(A_BB), bb = bb0 > 0 ? bb0 : 1.0;
    double rs = read_scalar(A_RR);  // ||r||^2 (the preconditioned product r.z drives the recurrences, not the stop test)

    // p . q as a by-product of the multiply where one window launch writes all of q (the rows' p sits in its LDS window): the
    // launch leaves one partial per workgroup in the slot the dot kernel would fill -- at most `grid` of them, the rest of the
    // slot stays zero, and the update kernel adds the slot up in the same fixed order as ever.
    const int xy_parts = P->cfg.cg_fused_dot != 2 ? spmv_xy_partials(P) : 0;
    const bool fused = xy_parts > 0 && xy_parts <= grid;
    if (fused) HIP_TRY(hipMemsetAsync(s + (size_t)A_PQ * kMaxGrid, 0, kMaxGrid * sizeof(double), st));
    auto enqueue_iteration = [&](int cur) -> int {
        // q = A p: x of the multiply changes every time
        const int e = fused ? spmv_xy(P, p, q, stream, s + (size_t)A_PQ * kMaxGrid) : ehyb_spmv(P, p, q, stream);
        if (e != EHYB_OK) return e;
        if (!fused) hipLaunchKernelGGL(cg_dot_kernel, dim3(grid), dim3(kThreads), 0, st, n, p, q, s);
        hipLaunchKernelGGL(cg_update_kernel, dim3(grid), dim3(kThreads), 0, st, n, p, q, dinv, x, r, s, cur);
        hipLaunchKernelGGL(cg_direction_kernel, dim3(grid), dim3(kThreads), 0, st, n, r, dinv, p, s, cur);
        return EHYB_OK;
    };
    // An even and an odd iteration, captured once and replayed: one submission per two iterations
    // instead of eight or ten launches.  cfg.graphs = 2 keeps the plain launches (A/B, debugging).
    if (P->cfg.graphs != 2 && max_iter >= 2 &&
        hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) == hipSuccess) {
        int erc = enqueue_iteration(0);
        if (erc == EHYB_OK) erc = enqueue_iteration(1);
        const hipError_t eend = hipStreamEndCapture(st, &W.graph);
        if (erc != EHYB_OK || eend != hipSuccess ||
            hipGraphInstantiate(&W.exec, W.graph, nullptr, nullptr, 0) != hipSuccess)
            W.exec = nullptr;
        (void)hipGetLastError();
    }
    int it = 0;
    while (it < max_iter && std::sqrt(rs / bb) > rtol) {
        const int burst = std::min(check_every, max_iter - it);  // even, except possibly the very last one
        int k = 0;
        for (; k + 2 <= burst; k += 2) {
            if (W.exec) {
                HIP_TRY(hipGraphLaunch(W.exec, st));
            } else {
                if ((rc = enqueue_iteration(0)) != EHYB_OK || (rc = enqueue_iteration(1)) != EHYB_OK) return rc;
            }
        }
        if (k < burst && (rc = enqueue_iteration(0)) != EHYB_OK) return rc;
        it += burst;
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(h.data(), s, h.size() * sizeof(double), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        rs = read_scalar(A_RR);
        const double rz = read_scalar(A_RZ0 + 2 * (burst & 1));
        if (!(rs == rs) || !(rz == rz)) {
            rs = NAN;
            break;  // NaN: breakdown (matrix or preconditioner not positive definite)
        }
    }
    if (iters_done) *iters_done = it;
    if (rel_residual) *rel_residual = std::sqrt(rs / bb);
    if (!(rs == rs)) EHYB_FAIL(EHYB_ERR_ARG, "ehyb_cg: breakdown (is the matrix symmetric positive definite?)");
    return EHYB_OK;
}

// ------------------------------------------------------------------ building blocks for a multi-GPU caller
// The same four vector kernels for a caller that owns the loop (ehyb_spmv_gpu_amd/dist.py HaloCG: one
// process per GPU, the multiply through the halo exchange): every rank runs them on its rows with the
// SAME grid, and an all_reduce (sum) of a slot of the partial array turns every rank's partials into
// the element-wise global ones -- the kernel that needs the scalar adds them up as before.
//   s: `slots` slots of `slot_doubles` doubles (ehyb_cg_layout); r.z number c (0/1) lives in slot rz0 + 2 c, r.r
//   between the two; every launch uses slot_doubles / 2 workgroups.
extern "C" int ehyb_cg_layout(int* slots, int* slot_doubles, int* slot_bb, int* slot_pq, int* slot_rr, int* slot_rz0)
{
    if (slots) *slots = A_COUNT;
    if (slot_doubles) *slot_doubles = kMaxGrid;
    if (slot_bb) *slot_bb = A_BB;
    if (slot_pq) *slot_pq = A_PQ;
    if (slot_rr) *slot_rr = A_RR;
    if (slot_rz0) *slot_rz0 = A_RZ0;
    return EHYB_OK;
}

static int check_vec(int n, const void* a, const void* b, const void* c, const char* who)
{
    if (n < 0 || !a || !b || !c) EHYB_FAIL(EHYB_ERR_ARG, "%s: bad arguments", who);
    return EHYB_OK;
}

// r = b - q, p = z = M^-1 r; partials of r.z (slot rz0), r.r, b.b
extern "C" int ehyb_cg_init_step(int n, const double* b, const double* q, const double* dinv, double* r, double* p, double* s,
                                 void* stream)
{
    int rc = check_vec(n, b, q, s, "ehyb_cg_init_step");
    if (rc != EHYB_OK || !r || !p) return rc != EHYB_OK ? rc : EHYB_ERR_ARG;
    hipLaunchKernelGGL(cg_init_kernel, dim3(kMaxGrid / 2), dim3(kThreads), 0, (hipStream_t)stream, n, b, q, dinv, r, p, s);
    HIP_TRY(hipGetLastError());
    return EHYB_OK;
}
// partials of p.q (slot pq)
extern "C" int ehyb_cg_dot_step(int n, const double* p, const double* q, double* s, void* stream)
{
    int rc = check_vec(n, p, q, s, "ehyb_cg_dot_step");
    if (rc != EHYB_OK) return rc;
    hipLaunchKernelGGL(cg_dot_kernel, dim3(kMaxGrid / 2), dim3(kThreads), 0, (hipStream_t)stream, n, p, q, s);
    HIP_TRY(hipGetLastError());
    return EHYB_OK;
}
// alpha = rz[cur] / pq; x += alpha p; r -= alpha q; partials of the new r.z (slot rz0 + (cur ^ 1)) and r.r
extern "C" int ehyb_cg_update_step(int n, const double* p, const double* q, const double* dinv, double* x, double* r, double* s,
                                   int cur, void* stream)
{
    int rc = check_vec(n, p, q, s, "ehyb_cg_update_step");
    if (rc != EHYB_OK || !x || !r) return rc != EHYB_OK ? rc : EHYB_ERR_ARG;
    hipLaunchKernelGGL(cg_update_kernel, dim3(kMaxGrid / 2), dim3(kThreads), 0, (hipStream_t)stream, n, p, q, dinv, x, r, s, cur & 1);
    HIP_TRY(hipGetLastError());
    return EHYB_OK;
}
// beta = rz[cur ^ 1] / rz[cur]; p = z + beta p
extern "C" int ehyb_cg_direction_step(int n, const double* r, const double* dinv, double* p, const double* s, int cur, void* stream)
{
    int rc = check_vec(n, r, p, s, "ehyb_cg_direction_step");
    if (rc != EHYB_OK) return rc;
    hipLaunchKernelGGL(cg_direction_kernel, dim3(kMaxGrid / 2), dim3(kThreads), 0, (hipStream_t)stream, n, r, dinv, p, s, cur & 1);
    HIP_TRY(hipGetLastError());
    return EHYB_OK;
}
