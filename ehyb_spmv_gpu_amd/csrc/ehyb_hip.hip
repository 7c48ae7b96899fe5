// HIP kernels for gfx950 (MI355X, CDNA4) and the device half of the C-ABI.
//
// Kernels (replace reference kernel.cu:43-284):
//   ehyb_ell_kernel  one workgroup per work item {partition, slab range}:
//                      1. stage the partition's x-window into LDS -- contiguous own segment
//                         (coalesced) + gathered halo columns (the "explicit cache",
//                         kernel.cu:137-141, grown to <= 160 KiB per workgroup);
//                      2. each wave64 walks 64-row slabs: per lane one row, per step one
//                         16-byte value pair (global_load_dwordx4, 1 KiB per wave) and one
//                         4-byte pair of 16-bit window-local columns, two LDS gathers
//                         (ds_read_b64) and two fp64 FMAs (kernel.cu:150-163);
//                      3. y[row] = dot, 512 B coalesced per slab.
//                    Static slab->wave assignment (slabs of a partition are sorted by width),
//                    so no work-queue atomics (kernel.cu:142,164-166) and nothing to reset.
//   ehyb_er_kernel   CSR residual: G lanes per segment (64/16/4 by segment length), strided
//                    coalesced (col,val) reads, x gathered from global memory (L2/MALL),
//                    wavefront shuffle reduction, y[row] += sum -- or one fp64 atomic per
//                    segment for rows split into several segments (the working form of
//                    kernel.cu:43-67 longRowKernel).  Runs on every multiply (the reference
//                    skips it after the first launch: SURVEY 8 a-10 item 1).
// No MFMA: 2 flops per 10-12 streamed bytes, HBM-bound.
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

// ------------------------------------------------------------------ window staging
// The LDS image of a partition's window starts at the even row at or below the partition start
// (the layout builder numbers window-local columns from there), so the contiguous part moves
// as 16-byte aligned double2 loads and ds_write_b128; up to four loads per thread are issued
// before the first LDS write.  Halo columns follow: index load, x gather, LDS write, again four
// deep.  win[0] may hold x[ps-1] when ps is odd; no entry refers to it.
template <int THREADS>
__device__ __forceinline__ void stage_window(double* __restrict__ win, const double* __restrict__ x, int ps,
                                             int wl, const int* __restrict__ hc, int hn)
{
    const int base = ps & ~1;
    const int cnt = wl + (ps & 1);
    const int n2 = cnt >> 1;
    const double2* __restrict__ x2 = reinterpret_cast<const double2*>(x + base);
    double2* __restrict__ w2 = reinterpret_cast<double2*>(win);
    for (int i0 = threadIdx.x; i0 < n2; i0 += 4 * THREADS) {
        double2 a[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (i0 + j * THREADS < n2) a[j] = x2[i0 + j * THREADS];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (i0 + j * THREADS < n2) w2[i0 + j * THREADS] = a[j];
    }
    if ((cnt & 1) && threadIdx.x == 0) win[cnt - 1] = x[base + cnt - 1];
    double* __restrict__ wh = win + cnt;
    for (int i0 = threadIdx.x; i0 < hn; i0 += 4 * THREADS) {
        int c[4];
        double a[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (i0 + j * THREADS < hn) c[j] = hc[i0 + j * THREADS];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (i0 + j * THREADS < hn) a[j] = x[c[j]];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (i0 + j * THREADS < hn) wh[i0 + j * THREADS] = a[j];
    }
}

// ------------------------------------------------------------------ residual tail
// The residual segments of a work item's rows, multiplied by the whole workgroup: G lanes per
// segment (64 / 16 / 4 by segment length, longest first), strided coalesced (col,val) reads, x
// gathered from global memory, shuffle reduction, then y[row] += sum -- plain for rows with one
// segment (rows are unique, kernel.cu:69-77), one fp64 atomic per segment for split rows (the
// working form of kernel.cu:43-67).  Called after the workgroup's own `y[row] = dot` stores
// and a __syncthreads() (fused), or from ehyb_er_kernel behind the ELL launch.
template <int G, int THREADS>
__device__ __forceinline__ void er_bin(int lo, int hi, const int64_t* __restrict__ seg_ptr,
                                       const int* __restrict__ seg_row, const int* __restrict__ col,
                                       const double* __restrict__ val, const double* __restrict__ x,
                                       double* __restrict__ y)
{
    constexpr int SEGS = THREADS / G;
    const int sub = threadIdx.x % G;
    for (int base = lo; base < hi; base += SEGS) {  // uniform trip count: every lane reaches the shuffles
        const int seg = base + threadIdx.x / G;
        double acc0 = 0.0, acc1 = 0.0;
        if (seg < hi) {
            const int64_t b = seg_ptr[seg], e = seg_ptr[seg + 1];
            int64_t k = b + sub;
            for (; k + G < e; k += 2 * G) {
                const int ca = col[k], cb = col[k + G];
                const double va = val[k], vb = val[k + G];
                acc0 = fma(va, x[ca], acc0);
                acc1 = fma(vb, x[cb], acc1);
            }
            if (k < e) acc0 = fma(val[k], x[col[k]], acc0);
        }
        double acc = acc0 + acc1;
#pragma unroll
        for (int off = G / 2; off > 0; off >>= 1) acc += __shfl_down(acc, off, G);
        if (sub == 0 && seg < hi) {
            const int r = seg_row[seg];
            if (r < 0)
                unsafeAtomicAdd(&y[r & 0x7fffffff], acc);
            else
                y[r] += acc;
        }
    }
}

template <int THREADS>
__device__ __forceinline__ void er_item(const int4 er, const int64_t* __restrict__ seg_ptr,
                                        const int* __restrict__ seg_row, const int* __restrict__ col,
                                        const double* __restrict__ val, const double* __restrict__ x,
                                        double* __restrict__ y)
{
    er_bin<64, THREADS>(er.x, er.y, seg_ptr, seg_row, col, val, x, y);
    er_bin<16, THREADS>(er.y, er.z, seg_ptr, seg_row, col, val, x, y);
    er_bin<4, THREADS>(er.z, er.w, seg_ptr, seg_row, col, val, x, y);
}

// ------------------------------------------------------------------ ELL kernel
// STAMP = true is a diagnostic instantiation (tools/ only): thread 0 of every workgroup records
// the 100 MHz wall clock at entry, after staging and at exit into a buffer of its own.
// EARLY = true shortens the cold-start chain of dependent misses (item -> partition arrays ->
// x / halo-index loads -> gathers -> barrier -> slab record -> lane map -> values: 7 hops): the
// partition scalars come with the item (item_part) and the first slab's record and lane map are
// requested before the window is staged, so they arrive while it is being filled.
template <int THREADS, bool STAMP = false, bool SCALAR_STAGE = false, bool FUSE_ER = false, bool EARLY = false>
__global__ __launch_bounds__(THREADS) void ehyb_ell_kernel(
    const int4* __restrict__ items, const int* __restrict__ part_boundary, const int* __restrict__ win_len,
    const int* __restrict__ halo_ptr, const int* __restrict__ halo_cols,
    const uint4* __restrict__ slab_meta, const uint8_t* __restrict__ lane_group,
    const double2* __restrict__ ell_val, const uint32_t* __restrict__ ell_col, const double* __restrict__ x,
    double* __restrict__ y, const int64_t* __restrict__ er_seg_ptr, const int* __restrict__ er_seg_row,
    const int* __restrict__ er_col, const double* __restrict__ er_val,
    unsigned long long* __restrict__ stamps = nullptr, const int4* __restrict__ item_part = nullptr)
{
    extern __shared__ __attribute__((aligned(16))) double win[];
    if (STAMP && threadIdx.x == 0) stamps[4 * blockIdx.x + 0] = wall_clock64();
    const int4 it = items[2 * blockIdx.x];
    const int p = it.x;
    int ps, pe, wl, hb, hn;
    if (EARLY) {
        const int4 a = item_part[2 * blockIdx.x], b = item_part[2 * blockIdx.x + 1];
        ps = a.x, pe = a.y, wl = a.z, hb = a.w, hn = b.x;
    } else {
        ps = part_boundary[p];
        pe = part_boundary[p + 1];
        wl = win_len[p];
        hb = halo_ptr[p];
        hn = halo_ptr[p + 1] - hb;
    }
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int WAVES = THREADS / 64;
    const int s_first = it.y + wave;
    uint4 sm_first = make_uint4(0, 0, 0, 0);
    uint32_t lg_first = 0;
    if (EARLY && s_first < it.z) {
        sm_first = slab_meta[s_first];
        lg_first = lane_group[(size_t)s_first * 64 + lane];
    }

    if (SCALAR_STAGE) {  // A/B arm: one dependent load per thread per pass
        const int base = ps & ~1, cnt = wl + (ps & 1);
        for (int i = threadIdx.x; i < cnt; i += THREADS) win[i] = x[base + i];
        for (int i = threadIdx.x; i < hn; i += THREADS) win[cnt + i] = x[halo_cols[hb + i]];
    } else {
        stage_window<THREADS>(win, x, ps, wl, halo_cols + hb, hn);
    }
    __syncthreads();
    if (STAMP && threadIdx.x == 0) stamps[4 * blockIdx.x + 1] = wall_clock64();

    for (int s = s_first; s < it.z; s += WAVES) {
        // slab record {first value pair, first column word, first row, pairs << 8 | groups - 1}
        const bool first = EARLY && s == s_first;
        const uint4 sm = first ? sm_first : slab_meta[s];
        const uint32_t lg = first ? lg_first : (uint32_t)lane_group[(size_t)s * 64 + lane];
        const int np = (int)(sm.w >> 8);
        const int G = (int)(sm.w & 0xffu) + 1;  // lanes with equal column lists share one word per pair
        const double2* __restrict__ v = ell_val + (size_t)sm.x * 64 + lane;
        const uint32_t* __restrict__ c = ell_col + sm.y + lg;
        double acc0 = 0.0, acc1 = 0.0;
        int k = 0;
        for (; k + 4 <= np; k += 4) {
            const double2 v0 = v[(k + 0) * 64], v1 = v[(k + 1) * 64], v2 = v[(k + 2) * 64], v3 = v[(k + 3) * 64];
            const uint32_t c0 = c[(k + 0) * G], c1 = c[(k + 1) * G], c2 = c[(k + 2) * G], c3 = c[(k + 3) * G];
            acc0 = fma(v0.x, win[c0 & 0xffffu], acc0);
            acc1 = fma(v0.y, win[c0 >> 16], acc1);
            acc0 = fma(v1.x, win[c1 & 0xffffu], acc0);
            acc1 = fma(v1.y, win[c1 >> 16], acc1);
            acc0 = fma(v2.x, win[c2 & 0xffffu], acc0);
            acc1 = fma(v2.y, win[c2 >> 16], acc1);
            acc0 = fma(v3.x, win[c3 & 0xffffu], acc0);
            acc1 = fma(v3.y, win[c3 >> 16], acc1);
        }
        for (; k < np; ++k) {
            const double2 v0 = v[k * 64];
            const uint32_t c0 = c[k * G];
            acc0 = fma(v0.x, win[c0 & 0xffffu], acc0);
            acc1 = fma(v0.y, win[c0 >> 16], acc1);
        }
        const int row = (int)sm.z + lane;
        if (row < pe) y[row] = acc0 + acc1;
    }
    if (FUSE_ER) {  // residual of this item's rows in the same launch (no second kernel boundary)
        const int4 er = items[2 * blockIdx.x + 1];
        if (er.w > er.x) {  // workgroup-uniform
            __syncthreads();  // the y stores above are complete and visible to the workgroup
            er_item<THREADS>(er, er_seg_ptr, er_seg_row, er_col, er_val, x, y);
        }
    }
    if (STAMP) {
        __syncthreads();
        if (threadIdx.x == 0) {
            stamps[4 * blockIdx.x + 2] = wall_clock64();
            unsigned xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            stamps[4 * blockIdx.x + 3] = xcc;
        }
    }
}

// ------------------------------------------------------------------ ELL kernel, pipelined
// Same arithmetic as ehyb_ell_kernel.  Differences, all about keeping HBM requests in flight:
//   * window staging issues 4 independent loads per thread before the first LDS write;
//   * the slab walk is one software-pipelined stream of 4-pair groups: the loads of group g+1
//     (possibly the first group of the wave's next slab) are issued before group g is
//     consumed, so a wave always has 4 x (1 KiB + 256 B) outstanding, also across slabs;
//   * a slab's last group is predicated (wave-uniform) instead of falling into a scalar tail.
struct EllGroup {
    double2 v[4];
    uint32_t c[4];
};

// Unconditional loads: pair indices past the slab's last pair are clamped onto it (same cache
// lines, no extra HBM traffic) and masked out when consumed, so the loop body stays one basic
// block and the compiler can count outstanding loads exactly.
__device__ __forceinline__ void ell_load(EllGroup& g, const double2* __restrict__ v,
                                         const uint32_t* __restrict__ c, int k, int last, int G)
{
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int idx = min(k + j, last);  // scalar
        g.v[j] = v[(size_t)idx * 64];
        g.c[j] = c[(size_t)idx * G];
    }
}

template <int THREADS, bool FUSE_ER = false>
__global__ __launch_bounds__(THREADS) void ehyb_ell_kernel_pipe(
    const int4* __restrict__ items, const int* __restrict__ part_boundary, const int* __restrict__ win_len,
    const int* __restrict__ halo_ptr, const int* __restrict__ halo_cols,
    const uint4* __restrict__ slab_meta, const uint8_t* __restrict__ lane_group,
    const double2* __restrict__ ell_val, const uint32_t* __restrict__ ell_col, const double* __restrict__ x,
    double* __restrict__ y, const int64_t* __restrict__ er_seg_ptr, const int* __restrict__ er_seg_row,
    const int* __restrict__ er_col, const double* __restrict__ er_val)
{
    extern __shared__ __attribute__((aligned(16))) double win[];
    const int4 it = items[2 * blockIdx.x];
    const int p = it.x;
    const int ps = part_boundary[p];
    const int pe = part_boundary[p + 1];
    const int wl = win_len[p];
    const int hb = halo_ptr[p];
    const int hn = halo_ptr[p + 1] - hb;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int WAVES = THREADS / 64;

    stage_window<THREADS>(win, x, ps, wl, halo_cols + hb, hn);
    __syncthreads();

    int s = it.y + wave;
    if (s < it.z) {
    uint4 sm = slab_meta[s];
    int np = (int)(sm.w >> 8);
    EllGroup ga, gb;  // ping-pong register sets: no copies, so no wait before the next issue
    ell_load(ga, ell_val + (size_t)sm.x * 64 + lane, ell_col + sm.y + lane_group[(size_t)s * 64 + lane], 0,
             max(np - 1, 0), (int)(sm.w & 0xffu) + 1);
    int row0 = (int)sm.z;

    double acc0 = 0.0, acc1 = 0.0;
    int k = 0;
    // One pipeline step: issue the loads of the group after CUR into NXT, then consume CUR.
#define ELL_STEP(CUR, NXT)                                                                        \
    {                                                                                             \
        const bool slab_end = k + 4 >= np;                                                        \
        const int ns = slab_end ? s + WAVES : s;                                                  \
        const bool has_next = ns < it.z;                                                          \
        const int ms = has_next ? ns : s; /* keep the prefetch addresses valid at the very end */ \
        const uint4 qm = slab_meta[ms];                                                           \
        const int nnp = (int)(qm.w >> 8);                                                         \
        const int nk = slab_end ? 0 : k + 4;                                                      \
        ell_load(NXT, ell_val + (size_t)qm.x * 64 + lane,                                         \
                 ell_col + qm.y + lane_group[(size_t)ms * 64 + lane], nk, max(nnp - 1, 0),        \
                 (int)(qm.w & 0xffu) + 1);                                                        \
        _Pragma("unroll") for (int j = 0; j < 4; ++j)                                             \
        {                                                                                         \
            const bool live = k + j < np; /* wave-uniform mask of the clamped pairs */            \
            const double vx = live ? CUR.v[j].x : 0.0;                                            \
            const double vy = live ? CUR.v[j].y : 0.0;                                            \
            acc0 = fma(vx, win[CUR.c[j] & 0xffffu], acc0);                                        \
            acc1 = fma(vy, win[CUR.c[j] >> 16], acc1);                                            \
        }                                                                                         \
        if (slab_end) {                                                                           \
            const int row = row0 + lane;                                                          \
            if (row < pe) y[row] = acc0 + acc1;                                                   \
            acc0 = 0.0;                                                                           \
            acc1 = 0.0;                                                                           \
            if (!has_next) break;                                                                 \
        }                                                                                         \
        s = ns;                                                                                   \
        k = nk;                                                                                   \
        np = nnp;                                                                                 \
        row0 = (int)qm.z;                                                                         \
    }
    for (;;) {
        ELL_STEP(ga, gb)
        ELL_STEP(gb, ga)
    }
#undef ELL_STEP
    }
    if (FUSE_ER) {
        const int4 er = items[2 * blockIdx.x + 1];
        if (er.w > er.x) {
            __syncthreads();
            er_item<THREADS>(er, er_seg_ptr, er_seg_row, er_col, er_val, x, y);
        }
    }
}

// ------------------------------------------------------------------ residual kernel
// Two-launch form (multi-GPU phase 2, or fuse_er = 2): one block per descriptor
// {seg_lo, seg_hi, lanes per segment}, a single pass of same-bin segments each.
template <int THREADS>
__global__ __launch_bounds__(THREADS) void ehyb_er_kernel(const int4* __restrict__ blocks,
                                                          const int64_t* __restrict__ seg_ptr,
                                                          const int* __restrict__ seg_row,
                                                          const int* __restrict__ col,
                                                          const double* __restrict__ val,
                                                          const double* __restrict__ x, double* __restrict__ y)
{
    const int4 b = blocks[blockIdx.x];
    if (b.z == 64)
        er_bin<64, THREADS>(b.x, b.y, seg_ptr, seg_row, col, val, x, y);
    else if (b.z == 16)
        er_bin<16, THREADS>(b.x, b.y, seg_ptr, seg_row, col, val, x, y);
    else
        er_bin<4, THREADS>(b.x, b.y, seg_ptr, seg_row, col, val, x, y);
}

// streaming-read probe for the on-box bandwidth ceiling
__global__ __launch_bounds__(256) void ehyb_read_kernel(const double2* __restrict__ src, size_t n2, double* sink)
{
    double acc = 0.0;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += stride) {
        double2 v = src[i];
        acc += v.x + v.y;
    }
    if (acc == 123.456) sink[0] = acc;  // keep the loads alive
}

// ------------------------------------------------------------------ launches
static int launch_ell(ehyb_plan* P, const double* x, double* y, hipStream_t st, bool fuse)
{
    const HostLayout& H = P->host;
    const int n_items = (int)(H.items.size() / 8);
    if (n_items == 0) return EHYB_OK;
    const size_t lds = (((size_t)H.lds_doubles * 8) + 15) / 16 * 16;
#define ELL_ARGS                                                                                            \
    (const int4*)P->d_items, P->d_part_boundary, P->d_win_len, P->d_halo_ptr, P->d_halo_cols,               \
        (const uint4*)P->d_slab_meta, P->d_lane_group, (const double2*)P->d_ell_val, P->d_ell_col, x, y,    \
        P->d_er_seg_ptr, P->d_er_seg_row, P->d_er_col, P->d_er_val
    const int var = P->cfg.ell_variant;  // 1 simple, 2 pipelined, 3 simple + scalar staging (A/B arms)
#define ELL_LAUNCH_F(T, F)                                                                                   \
    if (var == 2)                                                                                            \
        hipLaunchKernelGGL((ehyb_ell_kernel_pipe<T, F>), dim3(n_items), dim3(T), lds, st, ELL_ARGS);         \
    else if (var == 3)                                                                                       \
        hipLaunchKernelGGL((ehyb_ell_kernel<T, false, true, F>), dim3(n_items), dim3(T), lds, st, ELL_ARGS, nullptr); \
    else if (var == 4)                                                                                       \
        hipLaunchKernelGGL((ehyb_ell_kernel<T, false, true, F, true>), dim3(n_items), dim3(T), lds, st, ELL_ARGS,  \
                           nullptr, (const int4*)P->d_item_part);                                            \
    else                                                                                                     \
        hipLaunchKernelGGL((ehyb_ell_kernel<T, false, false, F>), dim3(n_items), dim3(T), lds, st, ELL_ARGS, nullptr);
#define ELL_LAUNCH(T)            \
    if (fuse) {                  \
        ELL_LAUNCH_F(T, true)    \
    } else {                     \
        ELL_LAUNCH_F(T, false)   \
    }
    switch (P->cfg.threads) {
        case 256: ELL_LAUNCH(256) break;
        case 512: ELL_LAUNCH(512) break;
        case 1024: ELL_LAUNCH(1024) break;
        default: EHYB_FAIL(EHYB_ERR_ARG, "ELL workgroup size %d not built (256/512/1024)", P->cfg.threads);
    }
#undef ELL_LAUNCH
#undef ELL_LAUNCH_F
#undef ELL_ARGS
    HIP_TRY(hipGetLastError());
    return EHYB_OK;
}

static int launch_er(ehyb_plan* P, const double* x, double* y, hipStream_t st)
{
    const HostLayout& H = P->host;
    if (H.er_bins[3] == 0) return EHYB_OK;
    const int n_blocks = (int)(H.er_blocks.size() / 4);
    if (P->cfg.er_threads != 256) EHYB_FAIL(EHYB_ERR_ARG, "residual workgroup size %d not built (256)", P->cfg.er_threads);
    hipLaunchKernelGGL(ehyb_er_kernel<256>, dim3(n_blocks), dim3(256), 0, st, (const int4*)P->d_er_blocks,
                       P->d_er_seg_ptr, P->d_er_seg_row, P->d_er_col, P->d_er_val, x, y);
    HIP_TRY(hipGetLastError());
    return EHYB_OK;
}

// Where the residual runs.  Two launches need a second ~8 us kernel boundary but give the residual
// thousands of independent blocks; the fused tail costs nothing when the residual is tiny and
// serialises it behind each workgroup's slabs when it is not (measured: tools/sweep.py --fuse).
// fuse_er: 1 = always fused, 2 = never, 0 = automatic: fused iff the residual holds < 0.2 % of
// the entries.  Multi-GPU plans keep the phases apart (phase 1 reads only the rank's x segment).
static bool fuse_residual(const ehyb_plan* P)
{
    if (P->cfg.n_top > 1 || P->cfg.fuse_er == 2) return false;
    if (P->cfg.fuse_er == 1) return true;
    const ehyb_stats& st = P->host.stats;
    return st.nnz_er * 500 < st.nnz;
}

template <class T>
static int upload(T** dst, const std::vector<T>& src)
{
    *dst = nullptr;
    size_t bytes = std::max<size_t>(src.size(), 1) * sizeof(T) + 4096;  // slack: clamped prefetches
    HIP_TRY(hipMalloc((void**)dst, bytes));
    if (!src.empty()) HIP_TRY(hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
    return EHYB_OK;
}

static void free_device(ehyb_plan* P)
{
    void* ptrs[] = {P->d_lane_group,    P->d_slab_meta,
                    P->d_part_boundary, P->d_win_len,     P->d_halo_ptr,  P->d_halo_cols, P->d_slab_pair_ptr,
                    P->d_slab_row,      P->d_ell_val,     P->d_ell_col,   P->d_items,     P->d_er_seg_ptr,
                    P->d_er_seg_row,    P->d_er_col,      P->d_er_val};
    for (void* q : ptrs)
        if (q) (void)hipFree(q);
    P->d_part_boundary = P->d_win_len = P->d_halo_ptr = P->d_halo_cols = nullptr;
    P->d_slab_pair_ptr = nullptr;
    P->d_slab_row = P->d_items = P->d_er_seg_row = P->d_er_col = nullptr;
    P->d_ell_val = P->d_er_val = nullptr;
    if (P->d_er_blocks) (void)hipFree(P->d_er_blocks);
    P->d_er_blocks = nullptr;
    if (P->d_item_part) (void)hipFree(P->d_item_part);
    P->d_item_part = nullptr;
    P->d_ell_col = nullptr;
    P->d_lane_group = nullptr;
    P->d_slab_meta = nullptr;
    P->d_er_seg_ptr = nullptr;
    P->uploaded = false;
}

extern "C" {

int ehyb_device_count(int* count)
{
    if (!count) EHYB_FAIL(EHYB_ERR_ARG, "ehyb_device_count: null");
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) {
        *count = 0;
        set_error("hipGetDeviceCount: %s", hipGetErrorString(e));
        return EHYB_ERR_NO_DEVICE;
    }
    *count = c;
    return EHYB_OK;
}

int ehyb_device_set(int device)
{
    HIP_TRY(hipSetDevice(device));
    return EHYB_OK;
}

int ehyb_device_name(char* buf, int len)
{
    if (!buf || len <= 0) EHYB_FAIL(EHYB_ERR_ARG, "ehyb_device_name: bad buffer");
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, dev));
    snprintf(buf, (size_t)len, "%s %s CUs=%d", prop.name, prop.gcnArchName, prop.multiProcessorCount);
    return EHYB_OK;
}

int ehyb_dev_alloc(size_t bytes, void** ptr)
{
    if (!ptr) EHYB_FAIL(EHYB_ERR_ARG, "ehyb_dev_alloc: null");
    HIP_TRY(hipMalloc(ptr, std::max<size_t>(bytes, 8)));
    return EHYB_OK;
}
int ehyb_dev_free(void* ptr)
{
    if (ptr) HIP_TRY(hipFree(ptr));
    return EHYB_OK;
}
int ehyb_h2d(void* dst, const void* src, size_t bytes)
{
    HIP_TRY(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
    return EHYB_OK;
}
int ehyb_d2h(void* dst, const void* src, size_t bytes)
{
    HIP_TRY(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
    return EHYB_OK;
}
int ehyb_dev_sync(void)
{
    HIP_TRY(hipDeviceSynchronize());
    return EHYB_OK;
}

int ehyb_measure_read_bw(size_t bytes, int iters, double* gbps)
{
    if (!gbps || iters < 1 || bytes < 4096) EHYB_FAIL(EHYB_ERR_ARG, "ehyb_measure_read_bw: bad arguments");
    double2* buf = nullptr;
    double* sink = nullptr;
    size_t n2 = bytes / sizeof(double2);
    HIP_TRY(hipMalloc((void**)&buf, n2 * sizeof(double2)));
    HIP_TRY(hipMalloc((void**)&sink, 8));
    HIP_TRY(hipMemset(buf, 0x11, n2 * sizeof(double2)));
    hipEvent_t a, b;
    HIP_TRY(hipEventCreate(&a));
    HIP_TRY(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(ehyb_read_kernel, dim3(4096), dim3(256), 0, 0, buf, n2, sink);
    HIP_TRY(hipEventRecord(a, 0));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(ehyb_read_kernel, dim3(4096), dim3(256), 0, 0, buf, n2, sink);
    HIP_TRY(hipEventRecord(b, 0));
    HIP_TRY(hipEventSynchronize(b));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, a, b));
    *gbps = (double)n2 * sizeof(double2) * iters / (ms * 1e-3) / 1e9;
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    (void)hipFree(buf);
    (void)hipFree(sink);
    return EHYB_OK;
}

// Diagnostic (tools/stamps.py): one launch of the stamped instantiation of the simple ELL kernel.
// out[4*i + {0,1,2,3}] = entry / staged / exit wall-clock ticks (100 MHz) and XCC id of item i.
int ehyb_debug_ell_stamps(ehyb_plan* P, const double* x, double* y, unsigned long long* out_host)
{
    if (!P || !P->uploaded || !out_host) EHYB_FAIL(EHYB_ERR_ARG, "ehyb_debug_ell_stamps: bad arguments");
    const HostLayout& H = P->host;
    const int n_items = (int)(H.items.size() / 8);
    const size_t lds = (((size_t)H.lds_doubles * 8) + 15) / 16 * 16;
    unsigned long long* d = nullptr;
    HIP_TRY(hipMalloc((void**)&d, (size_t)n_items * 32));
    HIP_TRY(hipMemset(d, 0, (size_t)n_items * 32));
#define STAMP_LAUNCH(T)                                                                                          \
    HIP_TRY(hipFuncSetAttribute((const void*)ehyb_ell_kernel<T, true>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                (int)lds));                                                                        \
    hipLaunchKernelGGL((ehyb_ell_kernel<T, true>), dim3(n_items), dim3(T), lds, 0, (const int4*)P->d_items,        \
                       P->d_part_boundary, P->d_win_len, P->d_halo_ptr, P->d_halo_cols,                           \
                       (const uint4*)P->d_slab_meta, P->d_lane_group, (const double2*)P->d_ell_val, P->d_ell_col, \
                       x, y, P->d_er_seg_ptr, P->d_er_seg_row, P->d_er_col, P->d_er_val, d);
    switch (P->cfg.threads) {
        case 256: STAMP_LAUNCH(256) break;
        case 512: STAMP_LAUNCH(512) break;
        case 1024: STAMP_LAUNCH(1024) break;
        default: EHYB_FAIL(EHYB_ERR_ARG, "workgroup size %d not built", P->cfg.threads);
    }
#undef STAMP_LAUNCH
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out_host, d, (size_t)n_items * 32, hipMemcpyDeviceToHost));
    (void)hipFree(d);
    return EHYB_OK;
}

int ehyb_plan_upload(ehyb_plan* P)
{
    clear_error();
    if (!P) EHYB_FAIL(EHYB_ERR_ARG, "ehyb_plan_upload: null plan");
    if (P->uploaded) return EHYB_OK;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count < 1)
        EHYB_FAIL(EHYB_ERR_NO_DEVICE, "ehyb_plan_upload: no HIP device visible (the EHYB multiply has no CPU fallback)");
    HIP_TRY(hipGetDevice(&P->device));
    const HostLayout& H = P->host;
    int rc;
#define UP(dst, src)                           \
    if ((rc = upload(&P->dst, H.src)) != EHYB_OK) { \
        free_device(P);                        \
        return rc;                             \
    }
    UP(d_part_boundary, part_boundary)
    UP(d_win_len, win_len)
    UP(d_halo_ptr, halo_ptr)
    UP(d_halo_cols, halo_cols)
    UP(d_slab_pair_ptr, slab_pair_ptr)
    UP(d_slab_row, slab_row)
    UP(d_ell_val, ell_val)
    UP(d_ell_col, ell_col)
    UP(d_lane_group, lane_group)
    UP(d_slab_meta, slab_meta)
    UP(d_items, items)
    UP(d_er_seg_ptr, er_seg_ptr)
    UP(d_er_seg_row, er_seg_row)
    UP(d_er_col, er_col)
    UP(d_er_val, er_val)
    UP(d_er_blocks, er_blocks)
    {
        // partition scalars per work item {first row, end row, contiguous window length, halo start}, {halo count}
        std::vector<int32_t> ip(H.items.size());
        for (size_t it = 0; it < H.items.size() / 8; ++it) {
            const int p = H.items[8 * it];
            ip[8 * it + 0] = H.part_boundary[p];
            ip[8 * it + 1] = H.part_boundary[p + 1];
            ip[8 * it + 2] = H.win_len[p];
            ip[8 * it + 3] = H.halo_ptr[p];
            ip[8 * it + 4] = H.halo_ptr[p + 1] - H.halo_ptr[p];
        }
        if ((rc = upload(&P->d_item_part, ip)) != EHYB_OK) {
            free_device(P);
            return rc;
        }
    }
#undef UP
    // opt in to the full 160 KiB of LDS (the role of cudaFuncSetAttribute at kernel.cu:351,411)
    const int lds = (int)((((size_t)H.lds_doubles * 8) + 15) / 16 * 16);
#define LDS_ATTR(K) HIP_TRY(hipFuncSetAttribute((const void*)(K), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
#define LDS_ATTR_T(T)                                     \
    LDS_ATTR((ehyb_ell_kernel<T, false, false, false>))   \
    LDS_ATTR((ehyb_ell_kernel<T, false, false, true>))    \
    LDS_ATTR((ehyb_ell_kernel<T, false, true, false>))    \
    LDS_ATTR((ehyb_ell_kernel<T, false, true, true>))     \
    LDS_ATTR((ehyb_ell_kernel<T, false, true, false, true>)) \
    LDS_ATTR((ehyb_ell_kernel<T, false, true, true, true>))  \
    LDS_ATTR((ehyb_ell_kernel_pipe<T, false>))            \
    LDS_ATTR((ehyb_ell_kernel_pipe<T, true>))
    LDS_ATTR_T(256)
    LDS_ATTR_T(512)
    LDS_ATTR_T(1024)
#undef LDS_ATTR_T
#undef LDS_ATTR
    P->uploaded = true;
    return EHYB_OK;
}

void ehyb_plan_destroy(ehyb_plan* P)
{
    if (!P) return;
    free_device(P);
    delete P;
}

int ehyb_spmv_phase(ehyb_plan* P, const double* x, double* y, void* stream, int phase)
{
    if (!P || !x || !y) EHYB_FAIL(EHYB_ERR_ARG, "ehyb_spmv: null argument");
    if (!P->uploaded) EHYB_FAIL(EHYB_ERR_STATE, "ehyb_spmv: plan not uploaded (no CPU fallback exists)");
    hipStream_t st = (hipStream_t)stream;
    int rc = EHYB_OK;
    if (phase == 0 && fuse_residual(P)) return launch_ell(P, x, y, st, true);  // one launch
    if (phase == 0 || phase == 1) rc = launch_ell(P, x, y, st, false);
    if (rc == EHYB_OK && (phase == 0 || phase == 2)) rc = launch_er(P, x, y, st);
    return rc;
}

int ehyb_spmv(ehyb_plan* P, const double* x, double* y, void* stream)
{
    return ehyb_spmv_phase(P, x, y, stream, 0);
}

int ehyb_spmv_bench(ehyb_plan* P, const double* x, double* y, void* stream, int warmup, int iters,
                    double* ms_total, double* ms_ell, double* ms_er)
{
    if (!P || iters < 1) EHYB_FAIL(EHYB_ERR_ARG, "ehyb_spmv_bench: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    int rc;
    for (int i = 0; i < warmup; ++i)
        if ((rc = ehyb_spmv(P, x, y, stream)) != EHYB_OK) return rc;
    hipEvent_t a, b;
    HIP_TRY(hipEventCreate(&a));
    HIP_TRY(hipEventCreate(&b));
    HIP_TRY(hipEventRecord(a, st));
    for (int i = 0; i < iters; ++i)
        if ((rc = ehyb_spmv(P, x, y, stream)) != EHYB_OK) return rc;
    HIP_TRY(hipEventRecord(b, st));
    HIP_TRY(hipEventSynchronize(b));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, a, b));
    if (ms_total) *ms_total = ms;
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    if (ms_ell || ms_er) {
        const int n = std::min(iters, 200);
        std::vector<hipEvent_t> ev((size_t)3 * n);
        for (auto& e : ev) HIP_TRY(hipEventCreate(&e));
        for (int i = 0; i < n; ++i) {
            const bool fused = fuse_residual(P);
            HIP_TRY(hipEventRecord(ev[3 * i + 0], st));
            if ((rc = launch_ell(P, x, y, st, fused)) != EHYB_OK) return rc;
            HIP_TRY(hipEventRecord(ev[3 * i + 1], st));
            if (!fused && (rc = launch_er(P, x, y, st)) != EHYB_OK) return rc;
            HIP_TRY(hipEventRecord(ev[3 * i + 2], st));
        }
        HIP_TRY(hipEventSynchronize(ev.back()));
        double se = 0, sr = 0;
        for (int i = 0; i < n; ++i) {
            float t1 = 0, t2 = 0;
            HIP_TRY(hipEventElapsedTime(&t1, ev[3 * i + 0], ev[3 * i + 1]));
            HIP_TRY(hipEventElapsedTime(&t2, ev[3 * i + 1], ev[3 * i + 2]));
            se += t1;
            sr += t2;
        }
        for (auto& e : ev) (void)hipEventDestroy(e);
        if (ms_ell) *ms_ell = se / n;
        if (ms_er) *ms_er = sr / n;
    }
    return EHYB_OK;
}

int ehyb_spmv_host(ehyb_plan* P, const double* x_host, double* y_host, int iters)
{
    if (!P || !x_host || !y_host || iters < 1) EHYB_FAIL(EHYB_ERR_ARG, "ehyb_spmv_host: bad arguments");
    if (!P->uploaded) EHYB_FAIL(EHYB_ERR_STATE, "ehyb_spmv_host: plan not uploaded (no CPU fallback exists)");
    const size_t n = (size_t)P->host.n_cols;
    double *dx = nullptr, *dy = nullptr;
    HIP_TRY(hipMalloc((void**)&dx, n * 8));
    HIP_TRY(hipMalloc((void**)&dy, n * 8));
    HIP_TRY(hipMemcpy(dx, x_host, n * 8, hipMemcpyHostToDevice));
    HIP_TRY(hipMemset(dy, 0, n * 8));
    int rc = EHYB_OK;
    for (int i = 0; i < iters && rc == EHYB_OK; ++i) rc = ehyb_spmv(P, dx, dy, nullptr);
    if (rc == EHYB_OK) {
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipMemcpy(y_host + P->host.row_begin, dy + P->host.row_begin,
                          (size_t)(P->host.row_end - P->host.row_begin) * 8, hipMemcpyDeviceToHost));
    }
    (void)hipFree(dx);
    (void)hipFree(dy);
    return rc;
}

int ehyb_plan_create(const matrixCOO* m, const ehyb_config* cfg, ehyb_plan** plan)
{
    if (!m) EHYB_FAIL(EHYB_ERR_ARG, "ehyb_plan_create: null matrix");
    int rc = ehyb_plan_create_host(m, 0, m->dimension, cfg, plan);
    if (rc != EHYB_OK) return rc;
    rc = ehyb_plan_upload(*plan);
    if (rc != EHYB_OK) {
        ehyb_plan_destroy(*plan);
        *plan = nullptr;
    }
    return rc;
}

// The drop-in entry point (reference spmv.cu:61-133).
int spmvGPuEHYB_status(matrixCOO* localMatrix, const double* vectorIn, double* vectorOut, const int MAXIter,
                       int* realIter)
{
    clear_error();
    if (!localMatrix || !vectorIn || !vectorOut || MAXIter < 0)
        EHYB_FAIL(EHYB_ERR_ARG, "spmvGPuEHYB: bad arguments");
    ehyb_config cfg;
    ehyb_config_default(&cfg);
    if (const char* v = getenv("EHYB_VERBOSE")) cfg.verbose = atoi(v);
    if (const char* v = getenv("EHYB_LDS_DOUBLES")) cfg.lds_doubles = atoi(v);
    if (const char* v = getenv("EHYB_THREADS")) cfg.threads = atoi(v);
    if (const char* v = getenv("EHYB_WINDOW_MODE")) cfg.window_mode = atoi(v);
    if (const char* v = getenv("EHYB_ITEMS_PER_CU")) cfg.items_per_cu = atoi(v);
    ehyb_plan* P = nullptr;
    int rc = ehyb_plan_create(localMatrix, &cfg, &P);  // COO2EHYB + upload (spmv.cu:73-81)
    if (rc != EHYB_OK) return rc;
    printf("sizeER is %lld\n", (long long)P->host.stats.size_er);  // spmv.cu:82
    const size_t n = (size_t)localMatrix->dimension;
    double *dx = nullptr, *dy = nullptr;
    auto fail = [&](int code) {
        if (dx) (void)hipFree(dx);
        if (dy) (void)hipFree(dy);
        ehyb_plan_destroy(P);
        return code;
    };
    if (hipMalloc((void**)&dx, n * 8) != hipSuccess || hipMalloc((void**)&dy, n * 8) != hipSuccess) {
        set_error("spmvGPuEHYB: device allocation of the vectors failed");
        return fail(EHYB_ERR_HIP);
    }
    if (hipMemcpy(dx, vectorIn, n * 8, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemset(dy, 0, n * 8) != hipSuccess) {
        set_error("spmvGPuEHYB: upload of x failed");
        return fail(EHYB_ERR_HIP);
    }
    double ms = 0;
    const int iters = std::max(1, MAXIter);
    rc = ehyb_spmv_bench(P, dx, dy, nullptr, 10, iters, &ms, nullptr, nullptr);  // spmv.cu:100-116
    if (rc != EHYB_OK) return fail(rc);
    if (hipMemcpy(vectorOut, dy, n * 8, hipMemcpyDeviceToHost) != hipSuccess) {
        set_error("spmvGPuEHYB: download of y failed");
        return fail(EHYB_ERR_HIP);
    }
    printf("iter is %d, time is %f ms, GPU Gflops is %f\n ", iters, ms,
           (1e-9 * ((double)localMatrix->totalNum * 2) * 1000 * iters) / ms);  // spmv.cu:121-122
    if (realIter) *realIter = iters;
    fail(EHYB_OK);
    return EHYB_OK;
}

void spmvGPuEHYB(matrixCOO* localMatrix, const double* vectorIn, double* vectorOut, const int MAXIter,
                 int* realIter)
{
    int rc = spmvGPuEHYB_status(localMatrix, vectorIn, vectorOut, MAXIter, realIter);
    if (rc != EHYB_OK) {
        fprintf(stderr, "spmvGPuEHYB failed (%d): %s\n", rc, ehyb_last_error());
        exit(rc);
    }
}

}  // extern "C"
