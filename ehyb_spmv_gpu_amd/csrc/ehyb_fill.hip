// Numeric phase of the EHYB build on the device.
//
// The reference builds every array of its format on one host thread and repeats all of it -- mt-metis,
// the scatter of I/J/V through the permutation (reordering.c:348-362), the fill of valBlockELL / valER
// (convert.c:316-369, 170-311) and a ~0.8 GB upload -- for every matrix it multiplies.  The layout of this
// library splits into a SYMBOLIC part (partition, windows, slabs, shared column words, pair orientation,
// work items: functions of the pattern alone, apart from the a_ij == a_ji test of symmetric pair storage)
// and a NUMERIC part: the value streams.  A plan built with cfg.value_map = 1 keeps, for every slot of
// every value stream, the entry of the source matrix it was filled from; the kernels below repeat the
// numeric part for new values on the same pattern (the next Newton step, the next time step, the next
// matrix of a parameter sweep) as one gather per stream, straight into the arrays the multiply reads:
//
//   ehyb_fill_kernel        dst[i] = src[i] < 0 ? 0.0 : V[order ? order[src[i]] : src[i]]
//                           ELL stream ([pair][lane][2] incl. inline residual pairs), CSR residual
//                           segments, panel stream of the residual -- whatever the plan holds on the device.
//                           `order` composes the caller's entry order before ehyb_matrix_reorder with the
//                           permuted one, so the V scatter of the reorder step is folded into the same gather.
//   ehyb_fill_check_kernel  before anything is written: every order[] value in range, and with symmetric
//                           pair storage V[a_ij] == V[a_ji] wherever one slot stands for both.
//
// HBM-bound and trivially so: 4 B (map) + 8 B gathered (the layout follows the permuted row order, so the
// gathers of a wave fall into a few lines) + 8 B written per slot.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <vector>

#include "ehyb_internal.h"

using namespace ehyb;

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) {                                                               \
            ::ehyb::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return _e == hipErrorNoDevice ? EHYB_ERR_NO_DEVICE : EHYB_ERR_HIP;                \
        }                                                                                     \
    } while (0)

namespace {

constexpr int kFillThreads = 256;
constexpr int kFillUnroll = 4;  // independent gathers in flight per lane

__global__ __launch_bounds__(kFillThreads) void ehyb_fill_kernel(double* __restrict__ dst, const int32_t* __restrict__ src, long long n,
                                                                 const double* __restrict__ V, const int32_t* __restrict__ order)
{
    const long long stride = (long long)gridDim.x * kFillThreads;
    for (long long base = (long long)blockIdx.x * kFillThreads + threadIdx.x; base < n; base += stride * kFillUnroll) {
        int32_t s[kFillUnroll];
        double v[kFillUnroll];
#pragma unroll
        for (int u = 0; u < kFillUnroll; ++u) {
            const long long i = base + u * stride;
            s[u] = i < n ? src[i] : -1;
        }
        if (order) {
#pragma unroll
            for (int u = 0; u < kFillUnroll; ++u)
                if (s[u] >= 0) s[u] = order[s[u]];
        }
#pragma unroll
        for (int u = 0; u < kFillUnroll; ++u) v[u] = s[u] >= 0 ? V[s[u]] : 0.0;
#pragma unroll
        for (int u = 0; u < kFillUnroll; ++u) {
            const long long i = base + u * stride;
            if (i < n) dst[i] = v[u];
        }
    }
}

// bad[0] += order values outside [0, count);  bad[1] += slots whose two entries differ (a_ij == a_ji is the test the layout builder paired them with)
__global__ __launch_bounds__(kFillThreads) void ehyb_fill_check_kernel(const int32_t* __restrict__ src, const int32_t* __restrict__ src2, long long n,
                                                                       const double* __restrict__ V, const int32_t* __restrict__ order, long long n_order,
                                                                       long long count, unsigned long long* __restrict__ bad)
{
    const long long stride = (long long)gridDim.x * kFillThreads;
    unsigned long long range = 0, pairs = 0;
    for (long long i = (long long)blockIdx.x * kFillThreads + threadIdx.x; i < n_order; i += stride)
        range += (unsigned long long)(long long)order[i] >= (unsigned long long)count;
    if (src2)
        for (long long i = (long long)blockIdx.x * kFillThreads + threadIdx.x; i < n; i += stride) {
            int32_t a = src[i], b = src2[i];
            if (a < 0 || b < 0) continue;
            if (order) {
                a = order[a], b = order[b];
                if ((unsigned long long)(long long)a >= (unsigned long long)count || (unsigned long long)(long long)b >= (unsigned long long)count) continue;  // counted above
            }
            pairs += !(V[a] == V[b]);  // the builder's own predicate (sym_orient_partition): +0.0 and -0.0 form a pair
        }
    if (range) atomicAdd(&bad[0], range);
    if (pairs) atomicAdd(&bad[1], pairs);
}

template <class T>
int to_device(T** dst, const T* src, size_t n)
{
    *dst = nullptr;
    HIP_TRY(hipMalloc((void**)dst, std::max<size_t>(n, 1) * sizeof(T)));
    if (n) {
        const hipError_t e = hipMemcpy(*dst, src, n * sizeof(T), hipMemcpyHostToDevice);
        if (e != hipSuccess) {  // never leave a half-filled array behind a non-null pointer
            (void)hipFree(*dst);
            *dst = nullptr;
            HIP_TRY(e);
        }
    }
    return EHYB_OK;
}

int grid_for(long long n)
{
    const long long per = (long long)kFillThreads * kFillUnroll;
    return (int)std::min<long long>(std::max<long long>((n + per - 1) / per, 1), 256 * 32);
}

struct Temp {  // host-array call: device copies that live for the call
    double* v = nullptr;
    int32_t* order = nullptr;
    unsigned long long* bad = nullptr;
    ~Temp()
    {
        if (v) (void)hipFree(v);
        if (order) (void)hipFree(order);
        if (bad) (void)hipFree(bad);
    }
};

}  // namespace

extern "C" int ehyb_plan_set_values(ehyb_plan* P, const double* values, int64_t count, const int32_t* entry_order, int on_device, void* stream)
{
    clear_error();
    if (!P || !values) EHYB_FAIL(EHYB_ERR_ARG, "ehyb_plan_set_values: null argument");
    if (!P->uploaded) EHYB_FAIL(EHYB_ERR_STATE, "ehyb_plan_set_values: plan not uploaded (the numeric phase runs on the device; no CPU fallback exists)");
    HostLayout& H = P->host;
    if (P->cfg.value_map != 1 || H.ell_src.size() != H.ell_val.size())
        EHYB_FAIL(EHYB_ERR_STATE, "ehyb_plan_set_values: the plan was built without cfg.value_map = 1 (or loaded from a cache file): it has no slot maps");
    if (count != H.src_entries)
        EHYB_FAIL(EHYB_ERR_ARG, "ehyb_plan_set_values: %lld values for a plan built from a matrix of %lld entries", (long long)count, (long long)H.src_entries);
    hipStream_t st = (hipStream_t)stream;
    int rc;
    // ---- slot maps to the device, once per plan
    if (!P->d_ell_src && (rc = to_device(&P->d_ell_src, H.ell_src.data(), H.ell_src.size())) != EHYB_OK) return rc;
    if (H.sym && !P->d_ell_src2 && (rc = to_device(&P->d_ell_src2, H.ell_src2.data(), H.ell_src2.size())) != EHYB_OK) return rc;
    if (P->d_er_val && !P->d_er_src && (rc = to_device(&P->d_er_src, H.er_src.data(), H.er_src.size())) != EHYB_OK) return rc;
    // (a panel form built on the device left its slot map there: H.pb_host_missing)
    if (P->d_pb_val && H.pb_host_missing && !P->d_pb_src) EHYB_FAIL(EHYB_ERR_STATE, "ehyb_plan_set_values: the device-built panel form kept no slot map");
    if (P->d_pb_val && !P->d_pb_src && (rc = to_device(&P->d_pb_src, H.pb_src.data(), H.pb_src.size())) != EHYB_OK) return rc;
    if ((P->d_er_val && H.er_src.size() != H.er_val.size()) || (P->d_pb_val && !H.pb_host_missing && H.pb_src.size() != H.pb_val.size()))
        EHYB_FAIL(EHYB_ERR_INTERNAL, "ehyb_plan_set_values: slot maps do not match the value streams");

    Temp T;
    const double* dV = values;
    const int32_t* dOrder = entry_order;
    if (!on_device) {
        if ((rc = to_device(&T.v, values, (size_t)count)) != EHYB_OK) return rc;
        dV = T.v;
        if (entry_order) {
            if ((rc = to_device(&T.order, entry_order, (size_t)count)) != EHYB_OK) return rc;
            dOrder = T.order;
        }
    }
    // ---- nothing is written before the input has been checked
    if (dOrder || H.sym) {
        HIP_TRY(hipMalloc((void**)&T.bad, 16));
        HIP_TRY(hipMemsetAsync(T.bad, 0, 16, st));
        const long long n = (long long)H.ell_src.size();
        hipLaunchKernelGGL(ehyb_fill_check_kernel, dim3(grid_for(std::max<long long>(n, count))), dim3(kFillThreads), 0, st, P->d_ell_src,
                           H.sym ? P->d_ell_src2 : nullptr, n, dV, dOrder, dOrder ? (long long)count : 0ll, (long long)count, T.bad);
        HIP_TRY(hipGetLastError());
        unsigned long long bad[2] = {0, 0};
        HIP_TRY(hipMemcpyAsync(bad, T.bad, 16, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        if (bad[0]) EHYB_FAIL(EHYB_ERR_ARG, "ehyb_plan_set_values: %llu entry_order values outside [0, %lld)", bad[0], (long long)count);
        if (bad[1])
            EHYB_FAIL(EHYB_ERR_ARG, "ehyb_plan_set_values: symmetric pair storage, but %llu stored pairs have a_ij != a_ji in the new values (plan unchanged)", bad[1]);
    }
    // ---- the gathers
    struct Job {
        double* dst;
        const int32_t* src;
        long long n;
    } jobs[3] = {{P->d_ell_val, P->d_ell_src, (long long)H.ell_src.size()},
                 {P->d_er_val, P->d_er_src, P->d_er_val ? (long long)H.er_src.size() : 0},
                 {P->d_pb_val, P->d_pb_src, P->d_pb_val ? (long long)H.pb_padded : 0}};
    for (const Job& j : jobs) {
        if (!j.dst || j.n == 0) continue;
        hipLaunchKernelGGL(ehyb_fill_kernel, dim3(grid_for(j.n)), dim3(kFillThreads), 0, st, j.dst, j.src, j.n, dV, dOrder);
        HIP_TRY(hipGetLastError());
    }
    P->host_values_stale = true;
    if (!on_device) HIP_TRY(hipStreamSynchronize(st));  // the temporaries go away with this call
    return EHYB_OK;
}
