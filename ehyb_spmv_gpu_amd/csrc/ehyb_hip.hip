// HIP kernels for gfx950 (MI355X, CDNA4) and the device half of the C-ABI.
//
// Kernels (replace reference kernel.cu:43-284):
//   ehyb_ell_kernel  one workgroup per work item = a run of 64-row slabs of (nearly) equal byte
//                    count, cut into segments at partition boundaries.  Per segment:
//                      1. stage the partition's x-window into LDS -- contiguous own segment
//                         (coalesced) + gathered halo columns (the "explicit cache",
//                         kernel.cu:137-141, grown to <= 160 KiB per workgroup);
//                      2. every wave64 takes slabs from an LDS counter (the reference's per-block
//                         work queue, kernel.cu:142,164-166 -- here re-armed per segment, so
//                         nothing survives a launch); per lane one row, per step one 16-byte
//                         value pair (global_load_dwordx4, 1 KiB per wave) and one 4-byte word
//                         holding two 16-bit window-local columns (shared by lanes with equal
//                         column lists), two LDS gathers (ds_read_b64) and two fp64 FMAs
//                         (kernel.cu:150-163);
//                      3. y[row] = dot, 512 B coalesced per slab.
//                    Two more forms of the same kernel:
//                      INLINE_ER  a tiny residual rides along as extra pairs behind every slab's
//                                 ELL pairs (global columns, x gathered from L2): one launch;
//                      SYM        symmetric pair storage: one workgroup per partition, the rows'
//                                 accumulators in LDS behind the x image; an entry marked in bit 15
//                                 of its column also adds value * x[own row] to row `column`
//                                 (ds_add_f64), so an in-partition pair a_ij == a_ji is read once.
//   ehyb_er_kernel   CSR residual: G lanes per segment (64/16/4 by segment length), strided
//                    coalesced (col,val) reads, x gathered from global memory (L2/MALL),
//                    wavefront shuffle reduction, y[row] += sum -- or one fp64 atomic per
//                    segment for rows split into several segments (the working form of
//                    kernel.cu:43-67 longRowKernel).  Runs on every multiply that has a residual
//                    not carried inline (the reference skips it after the first launch: SURVEY 8
//                    a-10 item 1); it is also phase 2 of the multi-GPU multiply (all remote columns).
// No MFMA: 2 flops per 5.8-10 streamed bytes, HBM-bound.
//
// Arms tried and dropped (measurements in DESIGN.md 3.1): software-pipelined slab walk with ping-pong
// register groups, 4-deep unrolled double2 staging, early slab-record loads, batched remainder
// pairs, global slab counters with work stealing, an 8-pair step for SYM.  The decisive levers were
// bytes (shared column lists, symmetric pairs) and scheduling (equal-cost items, one resident wave).
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

// ------------------------------------------------------------------ residual segments
// Residual segments [lo, hi) multiplied by one workgroup: G lanes per segment (64 / 16 / 4 by
// segment length, longest first), strided coalesced (col,val) reads, x gathered from global
// memory, shuffle reduction, then y[row] += sum -- plain for rows with one segment (rows are
// unique, kernel.cu:69-77), one fp64 atomic per segment for split rows (the working form of
// kernel.cu:43-67).  Called from ehyb_er_kernel, which runs behind the ELL launch.
// ASSIGN (direct shape, small matrices): y[row] = sum -- every row has exactly one segment.
// Most residual rows are short (R-MAT 2^22: 22 entries on average), so a lane has one to four entries
// and the time goes into the CHAIN of dependent loads, not into bandwidth: segment bounds -> (column,
// value) -> x[column] -> y.  Everything that does not depend on the products is therefore requested up
// front (row number and the old y with the bounds), and a lane's column/value loads are issued four at
// a time before the first gather of x (measured on R-MAT 2^22: DESIGN.md 3.2).
template <int G, int THREADS, bool ASSIGN>
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
        int r = 0;
        double y_old = 0.0;
        if (seg < hi) {
            const int64_t b = seg_ptr[seg], e = seg_ptr[seg + 1];
            r = seg_row[seg];
            if (!ASSIGN && sub == 0 && r >= 0) y_old = y[r];  // in flight while the products are formed
            int64_t k = b + sub;
            for (; k + 3 * G < e; k += 4 * G) {
                const int c0 = col[k], c1 = col[k + G], c2 = col[k + 2 * G], c3 = col[k + 3 * G];
                const double v0 = val[k], v1 = val[k + G], v2 = val[k + 2 * G], v3 = val[k + 3 * G];
                const double x0 = x[c0], x1 = x[c1], x2 = x[c2], x3 = x[c3];
                acc0 = fma(v0, x0, acc0);
                acc1 = fma(v1, x1, acc1);
                acc0 = fma(v2, x2, acc0);
                acc1 = fma(v3, x3, acc1);
            }
            // up to three more, again all requested before the first use
            const bool h0 = k < e, h1 = k + G < e, h2 = k + 2 * G < e;
            const int c0 = h0 ? col[k] : 0, c1 = h1 ? col[k + G] : 0, c2 = h2 ? col[k + 2 * G] : 0;
            const double v0 = h0 ? val[k] : 0.0, v1 = h1 ? val[k + G] : 0.0, v2 = h2 ? val[k + 2 * G] : 0.0;
            const double x0 = h0 ? x[c0] : 0.0, x1 = h1 ? x[c1] : 0.0, x2 = h2 ? x[c2] : 0.0;
            acc0 = fma(v0, x0, acc0);
            acc1 = fma(v1, x1, acc1);
            acc0 = fma(v2, x2, acc0);
        }
        double acc = acc0 + acc1;
#pragma unroll
        for (int off = G / 2; off > 0; off >>= 1) acc += __shfl_down(acc, off, G);
        if (sub == 0 && seg < hi) {
            if (ASSIGN)
                y[r] = acc;
            else if (r < 0)
                unsafeAtomicAdd(&y[r & 0x7fffffff], acc);
            else
                y[r] = y_old + acc;
        }
    }
}

// ------------------------------------------------------------------ ELL kernel
// items[2b]   = {first segment, end segment, -, -}      items[2b+1] = residual bins of the item
// segs[2g]    = {partition, first slab, end slab, halo count}
// segs[2g+1]  = {first row, end row, contiguous window length, halo start}
// DYN   waves take slabs from an LDS counter (the reference's per-block queue, kernel.cu:142,
//       164-166; here re-armed per segment) -- the default; false: slabs dealt round-robin (A/B arm).
//       A third arm -- global per-segment counters plus idle workgroups helping the busiest segment
//       -- was measured and dropped: the device-scope atomic per slab cost 6 % by itself and the
//       helping, at ~2 slabs per wave, evened the finish times without shortening the launch.
// STAMP:   diagnostic instantiation (tools/stamps.py only): thread 0 records the 100 MHz wall clock
//          at entry, after the first staging and at exit into a buffer of its own.
// INLINE_ER: slabs also multiply the residual pairs stored behind their ELL pairs (tiny residuals).
struct EllArgs {
    const int4* __restrict__ items;
    const int4* __restrict__ segs;
    const int* __restrict__ halo_cols;
    const uint4* __restrict__ slab_meta;
    const uint8_t* __restrict__ lane_group;
    const uint16_t* __restrict__ slab_lrow;  // SYM: the row (place in the LDS image) of every lane, 0xFFFF = none
    const double2* __restrict__ ell_val;
    const uint32_t* __restrict__ ell_col;
    const double* __restrict__ x;
    double* __restrict__ y;
    int win_cap;
    const int* __restrict__ item_map;  // non-null (ehyb_plan_tune): workgroup b takes item item_map[b]
    int xcd_map;  // 1: workgroup b takes item xcd_item(b), so that each XCD works on one contiguous run of items
    int windowless_zero;  // 1: a partition without a window gets y = 0 here; 0: the panel residual's second pass assigns its y
    unsigned long long* __restrict__ stamps;
    // non-null (ehyb_cg): the workgroup also leaves sum over its rows of y[row] * x[row] in xy_out[blockIdx.x] -- the p.q of
    // a conjugate-gradient step falls out of the multiply (the rows' x sits in the window, y in registers or accumulators)
    double* __restrict__ xy_out;
    // 1: the workgroup walks the slabs of a segment last to first.  Back-to-back multiplies of one plan alternate (cfg.ell_alternate):
    // what the previous launch streamed LAST is what still sits in the 256 MB Infinity Cache, and this launch reads it FIRST.
    int reverse;
    int reverse_items;  // with reverse, and more items than resident workgroups: workgroup b takes the items from the far end too
    // diagnostic launches only (stamps != null; ehyb_debug_ell_stamps_probe): > 0 = every window entry is staged from THREE vectors instead of
    // one (x and two shifted copies of it, `probe_n` entries long) -- what folding CG's direction update p = z + beta p into the staging
    // would gather (r, the old p, 1 / diag): how much longer the launch gets is the price of that fold
    int probe_n;
    // The value stream is read ONCE per multiply: loaded with the non-temporal hint it streams past the caches, which then hold what is read
    // again (column words shared by lanes, lane maps, x) -- 0.69 -> 0.78 of the peak for a launch that walks first to last, every entry stored
    // 1126 -> 1290 GFLOP/s (profiles/r04_nt_hints_ab.txt).  nt_slabs: the slabs at walk positions below nt_slabs/1024 of a segment are read
    // that way; the rest, the END of the walk, with plain loads -- what an alternating walk wants the Infinity Cache to keep for the next launch.
    int nt_slabs;
};

// Workgroups are handed to the 8 XCDs round robin (b mod 8).  With this map XCD k gets the k-th
// contiguous eighth of the items: neighbouring partitions, whose halo columns are each other's
// rows, then share one L2.
__device__ __forceinline__ int xcd_item(int b, int n)
{
    const int k = b & 7, j = b >> 3, chunk = n >> 3, rem = n & 7;
    return k * chunk + min(k, rem) + j;
}

// Which work item workgroup b takes: the tuned map of the plan (ehyb_plan_tune: the heaviest items on the XCDs that were
// measured fastest), else one contiguous run of items per XCD (plain storage), else item b.
__device__ __forceinline__ int item_of_block(const int* __restrict__ item_map, int xcd_map, int from_the_end = 0)
{
    const int b = from_the_end ? (int)gridDim.x - 1 - (int)blockIdx.x : (int)blockIdx.x;
    return item_map ? item_map[b] : (xcd_map ? xcd_item(b, (int)gridDim.x) : b);
}

// One entry of a slab: gather x from the window; SYM: bit 15 of the column says "this entry also
// stands for its mirror image": value * x[own row] goes to row `column`'s accumulator in LDS.
// The value of the next lane (lane + 1), 0 behind the last one: two DPP moves, no LDS traffic.
__device__ __forceinline__ double next_lane(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x130 /* wave_shl:1 */, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x130, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

// Lanes of a group (equal column lists: the unknowns of a node) send their mirror products to the
// SAME accumulator.  Summed across the lanes first (`code`: this lane adds for itself and the next
// 0/1/2 lanes, 3 = a lane before it adds for this one), a group of three costs one ds_add_f64
// instead of three that the hardware has to serialise.
#ifndef EHYB_SYM_GROUP_SUM
#define EHYB_SYM_GROUP_SUM 1
#endif
template <bool SYM>
__device__ __forceinline__ void ell_entry(double v, uint32_t col16, const double* __restrict__ win, double* yacc, double xi,
                                          int code, double& acc)
{
    if (SYM) {
        const uint32_t idx = col16 & 0x7fffu;
        acc = fma(v, win[idx], acc);
        if (EHYB_SYM_GROUP_SUM) {
            const double mine = (col16 & 0x8000u) ? v * xi : 0.0;
            const double n1 = next_lane(mine), n2 = next_lane(n1);
            const double sum = mine + ((code == 1 || code == 2) ? n1 : 0.0) + (code == 2 ? n2 : 0.0);
            if ((col16 & 0x8000u) && code != 3) unsafeAtomicAdd(&yacc[idx], sum);  // ds_add_f64
        } else {
            if (col16 & 0x8000u) unsafeAtomicAdd(&yacc[idx], v * xi);
        }
    } else {
        acc = fma(v, win[col16], acc);
    }
}

// the (value, value) pair of one lane: plain, or past the caches
template <bool NT>
__device__ __forceinline__ double2 ell_load_pair(const double2* __restrict__ p)
{
    if (NT) {
        double2 r;
        r.x = __builtin_nontemporal_load(&p->x);
        r.y = __builtin_nontemporal_load(&p->y);
        return r;
    }
    return *p;
}

template <bool INLINE_ER, bool SYM, bool NT>
__device__ __forceinline__ void ell_slab(const EllArgs& A, const double* __restrict__ win, double* yacc, int s, int base,
                                         int pe, int lane, double& xy)
{
    // slab record {first value pair, first column word, first row, pairs << 16 | residual pairs << 8 | groups - 1}
    const uint4 sm = A.slab_meta[s];
    const int np = (int)(sm.w >> 16);
    const int G = (int)(sm.w & 0x3fu) + 1;  // lanes with equal column lists share one word per pair
    const double2* __restrict__ v = A.ell_val + (size_t)sm.x * 64 + lane;
    // the lane's group (bits 0-5) and, with symmetric pairs, its part in the group's sum (bits 6-7)
    const uint32_t lgb = A.lane_group[(size_t)s * 64 + lane];
    const int code = SYM ? (int)(lgb >> 6) : 0;
    const uint32_t* __restrict__ c = A.ell_col + sm.y + (SYM ? (lgb & 0x3fu) : lgb);
    double acc0 = 0.0, acc1 = 0.0;
    if (INLINE_ER) {
        // Inline residual (tiny residuals only): `ner` more pairs behind the slab's ELL pairs, their
        // columns global -- [pair][2][lane] 32-bit words behind the slab's shared column words.
        // First, so that the loads are in flight while the ELL pairs stream; only the gather of x
        // from global memory (L2) waits for them.  No second launch, no read-modify-write of y.
        const int ner = (int)(sm.w >> 8) & 0xff;
        const double2* __restrict__ ve = v + (size_t)np * 64;
        const uint32_t* __restrict__ ce = A.ell_col + sm.y + (size_t)np * G + lane;
        for (int q = 0; q < ner; ++q) {
            const double2 vv = ve[q * 64];
            const uint32_t ca = ce[q * 128], cb = ce[q * 128 + 64];
            acc0 = fma(vv.x, A.x[ca], acc0);
            acc1 = fma(vv.y, A.x[cb], acc1);
        }
    }
    // Plain: lane l works on row sm.z + l.  SYM: the rows of a partition sit in its slabs longest
    // first (any order will do: sums go to the LDS accumulators by row), slab_lrow names the row.
    const int row = (int)sm.z + lane;
    const int lrow = SYM ? (int)A.slab_lrow[(size_t)s * 64 + lane] : row - base;  // place in the LDS image
    const bool has_row = SYM ? lrow != 0xFFFF : row < pe;
    const double xi = (SYM && has_row) ? win[lrow] : 0.0;
    // bit 7 of the record: the slab's columns are stored relative to the lane's own row (bands and
    // stencils: rows with equal offsets share their words); lanes without a row read column 0
    const uint32_t radd = (!SYM && (sm.w & 0x80u)) ? (uint32_t)lrow : 0u;
    const uint32_t cmask = (SYM || has_row) ? 0xffffu : 0u;
#define ELL_COL_LO(c) (SYM ? ((c) & 0xffffu) : ((((c) & 0xffffu) + radd) & cmask))
#define ELL_COL_HI(c) (SYM ? ((c) >> 16) : ((((c) >> 16) + radd) & cmask))
    int k = 0;
    // (an 8-pair step for SYM, 128 VGPRs at 16 waves per CU, measured 1 % slower than this one)
    for (; k + 4 <= np; k += 4) {
        const double2 v0 = ell_load_pair<NT>(v + (k + 0) * 64), v1 = ell_load_pair<NT>(v + (k + 1) * 64), v2 = ell_load_pair<NT>(v + (k + 2) * 64), v3 = ell_load_pair<NT>(v + (k + 3) * 64);
        const uint32_t c0 = c[(k + 0) * G], c1 = c[(k + 1) * G], c2 = c[(k + 2) * G], c3 = c[(k + 3) * G];
        ell_entry<SYM>(v0.x, ELL_COL_LO(c0), win, yacc, xi, code, acc0);
        ell_entry<SYM>(v0.y, ELL_COL_HI(c0), win, yacc, xi, code, acc1);
        ell_entry<SYM>(v1.x, ELL_COL_LO(c1), win, yacc, xi, code, acc0);
        ell_entry<SYM>(v1.y, ELL_COL_HI(c1), win, yacc, xi, code, acc1);
        ell_entry<SYM>(v2.x, ELL_COL_LO(c2), win, yacc, xi, code, acc0);
        ell_entry<SYM>(v2.y, ELL_COL_HI(c2), win, yacc, xi, code, acc1);
        ell_entry<SYM>(v3.x, ELL_COL_LO(c3), win, yacc, xi, code, acc0);
        ell_entry<SYM>(v3.y, ELL_COL_HI(c3), win, yacc, xi, code, acc1);
    }
    for (; k < np; ++k) {
        const double2 v0 = ell_load_pair<NT>(v + k * 64);
        const uint32_t c0 = c[k * G];
        ell_entry<SYM>(v0.x, ELL_COL_LO(c0), win, yacc, xi, code, acc0);
        ell_entry<SYM>(v0.y, ELL_COL_HI(c0), win, yacc, xi, code, acc1);
    }
#undef ELL_COL_LO
#undef ELL_COL_HI
    if (has_row) {
        if (SYM)
            unsafeAtomicAdd(&yacc[lrow], acc0 + acc1);  // other lanes scatter into the same accumulator
        else {
            A.y[row] = acc0 + acc1;
            if (A.xy_out != nullptr) xy = fma(acc0 + acc1, win[lrow], xy);  // (own rows are in the window)
        }
    }
}

// Stage the window of segment g and multiply its slabs.
// SYM (symmetric pair storage): the segment is a whole partition; its rows' accumulators sit in LDS
// right behind the x image, take the lanes' own sums and the scattered mirror products, and are
// written to y in one coalesced sweep at the end.
template <int THREADS, bool DYN, bool INLINE_ER, bool SYM, bool STAMP = false>
__device__ __forceinline__ void ell_segment(const EllArgs& A, double* __restrict__ win, int* __restrict__ next_slab,
                                            int g, int lane, int wave, double& xy)
{
    constexpr int WAVES = THREADS / 64;
    const int4 a = A.segs[2 * g], b = A.segs[2 * g + 1];
    const int sb = a.y, se = a.z, hn = a.w;
    const int ps = b.x, pe = b.y, wl = b.z, hb = b.w;
    if (!SYM && wl == 0 && hn == 0) {
        // a partition whose rows all went to the residual (its window did not pay, plan.cpp): nothing to
        // stage, no slab to walk -- the residual launch adds to y, so y = 0 in one coalesced sweep
        // (walking its empty slabs cost 22 us on R-MAT 2^22, 70 us on 2^24)
        if (!A.windowless_zero) return;  // pb_assign: pass 2 of the panel residual is the only writer of these rows
        const int r0 = max(ps, (int)A.slab_meta[sb].z), r1 = min(pe, r0 + (se - sb) * 64);
        for (int i = r0 + (int)threadIdx.x; i < r1; i += THREADS) A.y[i] = 0.0;
        return;
    }
    __syncthreads();  // every wave is done with the previous window and counter
    // The LDS image starts at the even row at or below the partition start (the layout builder
    // numbers window-local columns from there); win[0] may hold x[ps-1], unused.
    const int base = ps & ~1, cnt = wl + (ps & 1);
    double* yacc = win + cnt + hn;
    // (SYM: batching all of a thread's staging loads -- indices, then x, stores last -- measured no
    // faster than these loops: 6.1 vs 6.5 us of staging; the halo gathers set the pace)
    if (STAMP && A.probe_n > 0) {   // (diagnostic instantiation only: compiled out of the product's kernels)
        const int n = A.probe_n, o1 = n / 3, o2 = 2 * (n / 3);
        for (int i = threadIdx.x; i < cnt; i += THREADS) {
            const int c = min(base + i, n - 1);
            win[i] = A.x[c] + 1e-300 * (A.x[(c + o1) % n] + A.x[(c + o2) % n]);
        }
        for (int i = threadIdx.x; i < hn; i += THREADS) {
            const int c = A.halo_cols[hb + i];
            win[cnt + i] = A.x[c] + 1e-300 * (A.x[(c + o1) % n] + A.x[(c + o2) % n]);
        }
    } else if (STAMP && A.probe_n < 0) {   // (diagnostic: no halo gather at all -- results wrong, the launch span is what hiding the gather could reach at best)
        for (int i = threadIdx.x; i < cnt; i += THREADS) win[i] = A.x[base + i];
        for (int i = threadIdx.x; i < hn; i += THREADS) win[cnt + i] = 1.0;
    } else {
        for (int i = threadIdx.x; i < cnt; i += THREADS) win[i] = A.x[base + i];
        for (int i = threadIdx.x; i < hn; i += THREADS) win[cnt + i] = A.x[A.halo_cols[hb + i]];
    }
    if (SYM)
        for (int i = threadIdx.x; i < cnt; i += THREADS) yacc[i] = 0.0;
    if (DYN && threadIdx.x == 0) *next_slab = sb + WAVES;  // slabs sb..sb+WAVES-1 are pre-assigned
    __syncthreads();
    // diagnostic launches only (tools/stamps.py): when the first window of the item was staged
    if (A.stamps != nullptr && threadIdx.x == 0 &&
        g == A.items[2 * item_of_block(A.item_map, A.xcd_map)].x)
        A.stamps[4 * blockIdx.x + 1] = wall_clock64();
    int s = sb + wave;  // (logical position in the segment's walk; the slab it stands for depends on the direction)
    const int nt_end = sb + (int)(((long long)(se - sb) * A.nt_slabs + 1023) >> 10);   // walk positions below it: value stream past the caches
    while (s < se) {
        if (s < nt_end)
            ell_slab<INLINE_ER, SYM, true>(A, win, yacc, A.reverse ? se - 1 - (s - sb) : s, base, pe, lane, xy);
        else
            ell_slab<INLINE_ER, SYM, false>(A, win, yacc, A.reverse ? se - 1 - (s - sb) : s, base, pe, lane, xy);
        if (DYN) {
            int nx = 0;
            if (lane == 0) nx = atomicAdd(next_slab, 1);
            s = __builtin_amdgcn_readfirstlane(nx);
        } else {
            s += WAVES;
        }
    }
    if (SYM) {
        __syncthreads();  // all sums and scatters of the partition are in
        // (Folding the pairs that straddle two partitions as well -- 13 % fewer bytes on the bench
        // matrix -- would need y zeroed first and this write-out plus one add per halo column done
        // with global atomics: that alone was measured at +12.5 us per launch, more than the bytes save.)
        if (A.xy_out != nullptr) {
            for (int i = threadIdx.x + (ps & 1); i < cnt; i += THREADS) {
                A.y[base + i] = yacc[i];
                xy = fma(yacc[i], win[i], xy);
            }
        } else {
            for (int i = threadIdx.x + (ps & 1); i < cnt; i += THREADS) A.y[base + i] = yacc[i];
        }
    }
}

template <int THREADS, bool DYN, bool STAMP, bool INLINE_ER, bool SYM>
// Plain: <= 64 VGPRs (8 waves per SIMD), so that two 1024-thread workgroups share a CU when the caller picks
// a window of <= 80 KiB; at the default 160 KiB window one runs per CU (an 8-pair step with 128 VGPRs was
// measured there too: no gain).  SYM always runs one workgroup per CU (its window holds x and the y
// accumulators): 4 waves per SIMD, up to 128 VGPRs, no spills.
__global__ __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(SYM ? 4 : 8, 8))) void ehyb_ell_kernel(const EllArgs A)
{
    extern __shared__ __attribute__((aligned(16))) double win[];
    int* next_slab = reinterpret_cast<int*>(win + A.win_cap);  // one word behind the window
    if (STAMP && threadIdx.x == 0) A.stamps[4 * blockIdx.x + 0] = wall_clock64();
    const int4 it = A.items[2 * item_of_block(A.item_map, A.xcd_map, A.reverse_items)];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    double xy = 0.0;
    for (int sg = it.x; sg < it.y; ++sg) {
        ell_segment<THREADS, DYN, INLINE_ER, SYM, STAMP>(A, win, next_slab, sg, lane, wave, xy);
    }
    if (A.xy_out != nullptr) {  // (wave-uniform: a kernel argument)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) xy += __shfl_xor(xy, off, 64);
        __syncthreads();  // every wave is done with the last window
        if (lane == 0) win[wave] = xy;
        __syncthreads();
        if (threadIdx.x == 0) {
            double t = 0.0;
            for (int w = 0; w < THREADS / 64; ++w) t += win[w];  // fixed order
            A.xy_out[blockIdx.x] = t;
        }
    }
    if (STAMP) {
        __syncthreads();
        if (threadIdx.x == 0) {
            A.stamps[4 * blockIdx.x + 2] = wall_clock64();
            unsigned xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            A.stamps[4 * blockIdx.x + 3] = xcc;
        }
    }
}

// ------------------------------------------------------------------ residual kernel
// Two-launch form (multi-GPU phase 2, or a residual too large to ride in the ELL launch): one
// block per descriptor {seg_lo, seg_hi, lanes per segment}, a single pass of same-bin segments.
template <int THREADS, bool ASSIGN>
__global__ __launch_bounds__(THREADS) void ehyb_er_kernel(const int4* __restrict__ blocks,
                                                          const int64_t* __restrict__ seg_ptr,
                                                          const int* __restrict__ seg_row,
                                                          const int* __restrict__ col,
                                                          const double* __restrict__ val,
                                                          const double* __restrict__ x, double* __restrict__ y)
{
    const int4 b = blocks[blockIdx.x];  // (the XCD map of the ELL kernel was tried here: no difference on R-MAT)
    if (b.z == 64)
        er_bin<64, THREADS, ASSIGN>(b.x, b.y, seg_ptr, seg_row, col, val, x, y);
    else if (b.z == 16)
        er_bin<16, THREADS, ASSIGN>(b.x, b.y, seg_ptr, seg_row, col, val, x, y);
    else
        er_bin<4, THREADS, ASSIGN>(b.x, b.y, seg_ptr, seg_row, col, val, x, y);
}

// ------------------------------------------------------------------ panel residual (er_panel.cpp)
// Pass 1: one workgroup per unit {first column, columns, first entry, end entry}.  The unit's panel of
// x is staged in LDS; (value, 16-bit column word) are streamed, eight 64-entry chunks per wave and step.
// The column word carries two flags from which a lane works out the slot of its partial (er_panel.cpp,
// encode_panel_slots): bit 15 = first entry of a piece (entries of one row that are neighbours in the
// chunk), bit 14 = the piece's slot is not the previous piece's + 1: a "jump", with an entry in the jump list
// from which every lane behind it (up to the next jump) gets its slot by adding the pieces begun before its own.
// The products of a piece are summed in the wave's LDS words and its first lane stores the partial.
// One step of a segmented inclusive scan over the 64 lanes of a wave, in registers (DPP moves, no LDS): every lane takes
// (sum, flag) of the lane CTRL names -- row_shr:d inside the rows of 16 lanes, row_bcast15 / row_bcast31 across them --
// and adds the sum unless a piece has begun between that lane and itself (flag).  Lanes without a source read zeros.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ void seg_scan_step(double& v, uint32_t& f)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, true);
    const uint32_t fp = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)f, CTRL, ROW_MASK, 0xf, true);
    v += f ? 0.0 : __hiloint2double(hi, lo);
    f |= fp;
}

// Sums of the pieces of one 64-entry chunk: afterwards the LAST lane of every piece holds the piece's sum.  `heads` =
// ballot of the first lanes of the pieces (bit 0 always set).  Only the steps the chunk needs are run (wave-uniform
// branches on the ballot): none when every lane is its own piece -- most chunks of the sparse panels --, row_shr:1 alone
// when no piece is longer than two lanes, and so on; a hub row's 64-lane piece takes all six.
__device__ __forceinline__ double piece_sums(double prod, unsigned long long heads, bool head)
{
    const unsigned long long nh = ~heads;  // lanes that continue a piece
    if (nh == 0ull) return prod;
    if (heads == 1ull) {
        // the whole chunk is one piece -- a hub row in a hub panel, a third of the entries of a degree-ordered R-MAT: a
        // plain sum over the wave, no flags to carry (half the instructions of the segmented steps)
        double v = prod;
        v += __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x111, 0xf, 0xf, true), __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x111, 0xf, 0xf, true));
        v += __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x112, 0xf, 0xf, true), __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x112, 0xf, 0xf, true));
        v += __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x114, 0xf, 0xf, true), __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x114, 0xf, 0xf, true));
        v += __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x118, 0xf, 0xf, true), __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x118, 0xf, 0xf, true));
        v += __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x142, 0xa, 0xf, true), __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x142, 0xa, 0xf, true));
        v += __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x143, 0xc, 0xf, true), __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x143, 0xc, 0xf, true));
        return v;  // lane 63 holds the sum
    }
    double v = prod;
    uint32_t f = head ? 1u : 0u;
    seg_scan_step<0x111, 0xf>(v, f);  // row_shr:1
    const unsigned long long r2 = nh & (nh >> 1);
    if (r2 != 0ull) {  // a piece of three lanes or more
        seg_scan_step<0x112, 0xf>(v, f);  // row_shr:2
        const unsigned long long r4 = r2 & (r2 >> 2);
        if (r4 != 0ull) {  // five or more
            seg_scan_step<0x114, 0xf>(v, f);  // row_shr:4
            const unsigned long long r8 = r4 & (r4 >> 4);
            if (r8 != 0ull) seg_scan_step<0x118, 0xf>(v, f);  // nine or more: row_shr:8
        }
    }
    if (nh & 0x0001000100010000ull) {  // a piece crosses from one row of 16 lanes into the next
        seg_scan_step<0x142, 0xa>(v, f);  // row_bcast15: lane 15 -> row 1, lane 47 -> row 3
        seg_scan_step<0x143, 0xc>(v, f);  // row_bcast31: lane 31 -> rows 2 and 3
    }
    return v;
}

// SUMS_DPP (cfg.er_sums, the default): the products of a piece are added by the register scan above and the piece's
// LAST lane stores the partial; false = round 2's way: ds_add_f64 into 64 LDS words per wave, the FIRST lane reads the
// sum back and stores it (kept as the A/B arm: the LDS pipe of a CU was what bound pass 1 -- DESIGN.md 3.2).
// xcd_map: workgroup b takes unit xcd_item(b): the units of one panel (neighbours in the unit list) then run on ONE XCD at
// about the same time and stage their panel from its L2 instead of each from the fabric.
// PROBE: the timing-diagnostics instantiation (ehyb_debug_panel_times only); the product's own launches run PROBE = false,
// where every probe test folds away (they cost six vector instructions of ~70 per chunk).
// KCH = chunks per wave and step (8: 24 independent vector loads in flight per lane at 4 waves per SIMD).
// (Round 4 measured an instantiation for TWO 1024-thread workgroups per CU -- 9,728-column panels, KCH = 6, 59 VGPRs, 8 waves per SIMD, one
// workgroup staging while the other streams: R-MAT 2^22 140 against 132 us, 2^24 665 against 574 us, profiles/r04_d_panel_two_ab.jsonl -- the
// narrower panels' extra partial sums cost more than the occupancy gives; pass 1 alone ran level.  Removed again.)
template <int THREADS, bool SUMS_DPP, bool PROBE, int KCH>
__device__ __forceinline__ void pb_scale_body(const int2* __restrict__ items, const int4* __restrict__ units,
                                                                const double* __restrict__ val,
                                                                const uint16_t* __restrict__ colf,
                                                                const uint32_t* __restrict__ chunk,
                                                                const uint32_t* __restrict__ jump,
                                                                const double* __restrict__ x,
                                                                double* __restrict__ partial, int panel_cols, int probe_arg, int xcd_map,
                                                                int* __restrict__ queue, int n_items, int reverse)
{
    const int probe = PROBE ? probe_arg : 0;
    // probe (tools/panel_sweep.py, timing diagnostics only, results wrong): 1 no lane sums, 2 no stores,
    // 4 no LDS gather, 8 no panel staging
    extern __shared__ __attribute__((aligned(16))) double win[];
    constexpr int WAVES = THREADS / 64;
    // An ITEM = a run of units of (nearly) equal total cost, cut by the host (er_panel.cpp); every unit is a stretch of one
    // panel's entries and stages that panel once.
    // queue == null: one workgroup per item (workgroup b takes item xcd_item(b) / b).
    // queue != null (cfg.er_queue = 1, an A/B arm -- see DESIGN.md 3.2): one RESIDENT round of workgroups, each taking items until none is left.  The
    // hardware deals workgroups to the 8 XCDs round robin, so with one item per workgroup every XCD gets an eighth of the
    // work whatever its speed -- and two of the eight XCDs of every box measured stream 8-12 % slower than the fastest, which
    // the whole launch then waits for.  Here XCD k's workgroups take the items of the k-th contiguous eighth (queue[16 k] =
    // items taken: the units of one panel still meet in one L2), and a workgroup whose own eighth is used up takes from the
    // eighth with the most items left.  Exit: every workgroup leaves when every queue is empty (counts only grow); the last
    // one to leave (queue[128] = workgroups gone) zeroes the counts for the next launch.
    // (the item handed from thread 0 to the workgroup: one word behind the panel and the piece accumulators, in the dynamic
    // allocation -- a static __shared__ word on top of a 160 KiB dynamic limit is refused by hipFuncSetAttribute)
    int& s_item = *reinterpret_cast<int*>(win + panel_cols + (SUMS_DPP ? 0 : THREADS));
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    double* scr = win + panel_cols + 64 * wave;  // this wave's 64 piece accumulators, behind the panel (!SUMS_DPP only)
    if (!SUMS_DPP) scr[lane] = 0.0;
    int staged_x = -1, staged_n = -1;  // the panel in this workgroup's LDS (first column, columns): wave-uniform
    int my_q = 0;
    if (queue != nullptr && threadIdx.x == 0) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        my_q = (int)(xcc & 7u);
    }
  for (;;) {
    int2 it;
    if (queue != nullptr) {
        __syncthreads();  // every wave is done with the previous item's panel, and with s_item
        if (threadIdx.x == 0) {
            int item = -1;
            for (;;) {
                const int first = (int)((long long)n_items * my_q / 8), len = (int)((long long)n_items * (my_q + 1) / 8) - first;
                const int idx = len > 0 ? atomicAdd(&queue[16 * my_q], 1) : len;
                if (idx < len) {
                    item = reverse ? first + len - 1 - idx : first + idx;   // (alternating walk: every eighth from its far end)
                    break;
                }
                int best = -1, most = 0;  // own eighth used up: the one with the most items left
                for (int k = 0; k < 8; ++k) {
                    const int lk = (int)((long long)n_items * (k + 1) / 8) - (int)((long long)n_items * k / 8);
                    const int left = lk - __hip_atomic_load(&queue[16 * k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (left > most) most = left, best = k;
                }
                if (best < 0) break;  // nothing left anywhere
                my_q = best;
            }
            s_item = item;
        }
        __syncthreads();
        const int item = s_item;
        if (item < 0) break;
        it = items[item];
    } else {
        // reverse (successive launches alternate, as the ELL launch does): the items last to first, the units of an item last to
        // first, a unit's chunks last to first -- this launch starts with what the one before it left in the Infinity Cache
        const int idx = xcd_map ? xcd_item(blockIdx.x, gridDim.x) : (int)blockIdx.x;
        it = items[reverse ? (int)gridDim.x - 1 - idx : idx];
    }
  for (int ui = it.x; ui < it.y; ++ui) {
    const int un = reverse ? it.y - 1 - (ui - it.x) : ui;
    const int4 u = units[un];
    // the panel this workgroup staged last is still in its LDS: a unit of the same panel (the next stretch of a hub panel's
    // entries -- with the work queues a workgroup takes neighbouring items) streams straight away
    const bool staged_already = u.x == staged_x && u.y == staged_n;
    if (!staged_already && (ui != it.x || staged_n >= 0)) __syncthreads();  // every wave is done with the previous panel
    staged_x = u.x, staged_n = u.y;
    // stage the panel: all of a thread's loads in flight before the first store (a 64 KiB panel is 16
    // double2 loads per thread; one load per loop trip would pay the memory latency 16 times)
    if (!(probe & 8) && !staged_already) {
        const double2* __restrict__ xp2 = reinterpret_cast<const double2*>(x + u.x);  // panels start on even columns
        double2* win2 = reinterpret_cast<double2*>(win);
        const int n2 = u.y >> 1;
        for (int i0 = 0; i0 < n2; i0 += 8 * THREADS) {
            double2 t[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int i = i0 + j * THREADS + (int)threadIdx.x;
                t[j] = i < n2 ? xp2[i] : double2{0.0, 0.0};
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int i = i0 + j * THREADS + (int)threadIdx.x;
                if (i < n2) win2[i] = t[j];
            }
        }
        if ((u.y & 1) && threadIdx.x == 0) win[u.y - 1] = x[u.x + u.y - 1];
    }
    if (!staged_already) __syncthreads();
    const int c0 = u.z >> 6, c1 = u.w >> 6;  // chunks of 64 entries
    constexpr int K = KCH;  // chunks per wave and step: 24 independent vector loads in flight per lane at K = 8
    // The jump-list range of a chunk is known from the chunk records alone (wave-uniform, scalar loads): they are
    // fetched one step ahead, so that the jump entries travel together with the values and column words instead
    // of behind them (a gather that waits for the flags doubled the latency per step: 345 -> 470 us on R-MAT 2^24).
    uint32_t f0[K], fn[K];
    // the wave's steps: chunks c0 + K (wave + t WAVES) .., t = 0 .. steps - 1, walked up or down
    const int first = c0 + K * wave;
    const int steps = first < c1 ? (c1 - first + K * WAVES - 1) / (K * WAVES) : 0;
    const int dc = reverse ? -K * WAVES : K * WAVES;
    const int cstart = reverse ? first + (steps - 1) * K * WAVES : first;
#pragma unroll
    for (int j = 0; j < K; ++j) {
        const int cj = max(c0, min(cstart + j, c1 - 1));
        f0[j] = chunk[cj];
        fn[j] = chunk[cj + 1] - f0[j];
    }
    int c = cstart;
    for (int t = 0; t < steps; ++t, c += dc) {
        double v[K];
        uint32_t cw[K], jv[K];
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const int cj = c + j < c1 ? c + j : c;  // wave-uniform
            const size_t pos = (size_t)cj * 64 + lane;
            // (streamed past the caches: every entry is read once per multiply, and what the caches hold instead -- the x panels the units of a
            // hub panel stage again and again, the partial sums pass 2 is about to read -- is read again.  R-MAT 2^22: 128.0 -> 124.0 us)
            v[j] = __builtin_nontemporal_load(&val[pos]);
            cw[j] = __builtin_nontemporal_load(&colf[pos]);
            jv[j] = (uint32_t)lane < fn[j] ? jump[f0[j] + lane] : 0u;  // lane l: the chunk's l-th jump entry
        }
        uint32_t g0[K], gn[K];  // the records of the next step
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const int cj = max(c0, min(c + dc + j, c1 - 1));
            g0[j] = chunk[cj];
            gn[j] = chunk[cj + 1] - g0[j];
        }
        uint32_t slot[K], piece[K];
        unsigned long long hd[K];
        double xw[K];
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const bool head = (cw[j] & 0x8000u) != 0, jmp = (cw[j] & 0x4000u) != 0;
            const unsigned long long heads = __ballot(head), jumps = __ballot(jmp);
            // pieces / jumps begun in the lanes below this one (v_mbcnt), plus its own
            const uint32_t hc = __builtin_amdgcn_mbcnt_hi((uint32_t)(heads >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)heads, 0u)) + (head ? 1u : 0u);
            const uint32_t jc = __builtin_amdgcn_mbcnt_hi((uint32_t)(jumps >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)jumps, 0u)) + (jmp ? 1u : 0u);
            // the entry of the last jump at or below this lane sits in lane jc - 1 (lane 0 is always a jump)
            const uint32_t base = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((jc - 1u) << 2), (int)jv[j]);
            piece[j] = hc - 1u;
            // who stores the partial: the piece's last lane (register scan) or its first (LDS sums); every lane of a piece
            // computes the same slot.  0xFFFFFFFF: nothing to store (also what the padding piece yields)
            const bool stores = SUMS_DPP ? (lane == 63 || ((heads >> (lane + 1)) & 1ull)) : head;
            slot[j] = stores ? base + hc - 1u : 0xFFFFFFFFu;
            hd[j] = heads;
            const uint32_t cl = cw[j] & 0x3FFFu;
            xw[j] = (probe & 4) ? (double)cl : win[cl];
            cw[j] = (heads == ~0ull ? 1u : 0u) | (head ? 2u : 0u);  // bit 0: every lane its own piece, no sums needed
        }
#pragma unroll
        for (int j = 0; j < K; ++j) {
            if (c + j < c1) {  // wave-uniform
                const double prod = v[j] * xw[j];
                double sum = prod;
                if (SUMS_DPP) {
                    if (!(probe & 1)) sum = piece_sums(prod, hd[j], (cw[j] & 2u) != 0);
                } else if (!(cw[j] & 1u) && !(probe & 1)) {
                    // Some lanes share a slot: the piece sums are formed in this wave's 64 LDS words (zero between
                    // uses), one ds_add_f64 per lane, one read + one store of zero per piece.  (Shuffle trees --
                    // six ds_bpermute rounds per chunk -- cost 17 us of a 127 us launch here and 17 of 80 in pass 2.)
                    unsafeAtomicAdd(&scr[piece[j]], prod);
                    if (slot[j] != 0xFFFFFFFFu) sum = scr[piece[j]];
                    __builtin_amdgcn_wave_barrier();
                    scr[piece[j]] = 0.0;
                }
                if (slot[j] != 0xFFFFFFFFu && (!(probe & 2) || sum == 123.456)) {
                    if (probe & 256)
                        __builtin_nontemporal_store(sum, &partial[slot[j]]);
                    else
                        partial[slot[j]] = sum;
                }
            }
        }
#pragma unroll
        for (int j = 0; j < K; ++j) f0[j] = g0[j], fn[j] = gn[j];
    }
  }
    if (queue == nullptr) break;
  }
    if (queue != nullptr && threadIdx.x == 0 && atomicAdd(&queue[128], 1) == (int)gridDim.x - 1) {
        for (int k = 0; k < 8; ++k) __hip_atomic_store(&queue[16 * k], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&queue[128], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

template <int THREADS, bool SUMS_DPP, bool PROBE>
__global__ __launch_bounds__(THREADS) void ehyb_pb_scale_kernel(const int2* __restrict__ items, const int4* __restrict__ units, const double* __restrict__ val,
                                                                const uint16_t* __restrict__ colf, const uint32_t* __restrict__ chunk,
                                                                const uint32_t* __restrict__ jump, const double* __restrict__ x, double* __restrict__ partial,
                                                                int panel_cols, int probe_arg, int xcd_map, int* __restrict__ queue, int n_items, int reverse)
{
    pb_scale_body<THREADS, SUMS_DPP, PROBE, 8>(items, units, val, colf, chunk, jump, x, partial, panel_cols, probe_arg, xcd_map, queue, n_items, reverse);
}

// Pass 2: one workgroup per unit {first partial, end partial, first row, rows}.  The row block's
// accumulators live in LDS; (partial, 16-bit local row) are streamed and added (ds_add_f64); finally
// y[row] += accumulator for the rows that received something (the ELL launch has written y before) -- or,
// for a block of rows whose partitions have no window (rows < 0 in the unit), y[row] = accumulator for
// every row: the ELL launch leaves those rows alone.
// NT: the partial sums and their row words are streamed past the caches -- where they are more than the Infinity Cache can hold between the
// passes anyway (R-MAT 2^24: 460 MB; 520 -> 487 us with it, because the next multiply then finds more of the entry stream's tail there);
// where they fit (2^22: 89 MB) pass 2 reads them from that cache and the hint costs 3 us.
template <int THREADS, bool NT>
__global__ __launch_bounds__(THREADS) void ehyb_pb_reduce_kernel(const int4* __restrict__ units,
                                                                 const double* __restrict__ partial,
                                                                 const uint16_t* __restrict__ row,
                                                                 double* __restrict__ y, int probe)
{
    // probe (timing diagnostics only): 16 no lane sums, 32 no LDS adds, 64 no write-back, 128 no zeroing
    extern __shared__ __attribute__((aligned(16))) double yacc[];
    int4 u = units[blockIdx.x];
    const bool assign = u.w < 0;  // the block is the only writer of its rows (partitions without a window): y = sum, zeros included
    u.w = assign ? -u.w : u.w;
    if (!(probe & 128))
        for (int i = threadIdx.x; i < u.w; i += THREADS) yacc[i] = 0.0;
    __syncthreads();
    constexpr int K = 8;  // partials per thread and step: 16 independent loads in flight
    // every wave runs the same number of steps (the shuffles need all 64 lanes)
    for (int base = u.x; base < u.y; base += K * THREADS) {
        double v[K];
        uint32_t r[K];
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const int i = base + j * THREADS + (int)threadIdx.x;
            const bool in = i < u.y;
            if (NT) {
                v[j] = in ? __builtin_nontemporal_load(&partial[i]) : 0.0;
                r[j] = in ? (uint32_t)__builtin_nontemporal_load(&row[i]) : 0xFFFFFFFFu;
            } else {
                v[j] = in ? partial[i] : 0.0;
                r[j] = in ? (uint32_t)row[i] : 0xFFFFFFFFu;
            }
        }
#pragma unroll
        for (int j = 0; j < K; ++j) {
            // (summing equal neighbouring rows across lanes first was measured: 17 us of an 80 us launch
            // for nothing -- the LDS adds serialise the few same-row neighbours by themselves)
            if (r[j] != 0xFFFFFFFFu && (!(probe & 32) || v[j] == 123.456)) unsafeAtomicAdd(&yacc[r[j]], v[j]);  // ds_add_f64
        }
    }
    __syncthreads();
    // y[row] += accumulator for the rows that received something: the loads of a batch first, then the stores
    double* __restrict__ yp = y + u.z;
    if (probe & 64) return;
    if (assign) {
        for (int i = threadIdx.x; i < u.w; i += THREADS) yp[i] = yacc[i];
        return;
    }
    for (int i0 = 0; i0 < u.w; i0 += K * THREADS) {
        double a[K], yo[K];
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const int i = i0 + j * THREADS + (int)threadIdx.x;
            a[j] = i < u.w ? yacc[i] : 0.0;
        }
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const int i = i0 + j * THREADS + (int)threadIdx.x;
            yo[j] = a[j] != 0.0 ? yp[i] : 0.0;
        }
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const int i = i0 + j * THREADS + (int)threadIdx.x;
            if (a[j] != 0.0) yp[i] = yo[j] + a[j];
        }
    }
}

// dst[i] = src[idx[i]]: the send list of a halo exchange (multi-GPU), four gathers per thread in flight
__global__ __launch_bounds__(256) void ehyb_gather_kernel(const double* __restrict__ src, const int32_t* __restrict__ idx, double* __restrict__ dst, long long n)
{
    const long long base = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
    int32_t k[4];
    double v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) k[j] = base + j < n ? idx[base + j] : 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = base + j < n ? src[k[j]] : 0.0;
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (base + j < n) dst[base + j] = v[j];
}

// y[idx[i]] += src[i]: partial sums computed elsewhere, added into this rank's rows (an index may occur more than once)
__global__ __launch_bounds__(256) void ehyb_scatter_add_kernel(double* __restrict__ y, const int32_t* __restrict__ idx, const double* __restrict__ src, long long n)
{
    const long long base = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
    int32_t k[4];
    double v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) k[j] = base + j < n ? idx[base + j] : 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = base + j < n ? src[base + j] : 0.0;
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (base + j < n) unsafeAtomicAdd(&y[k[j]], v[j]);
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
static size_t ell_lds_bytes(const HostLayout& H) { return ((size_t)H.lds_doubles + 1) / 2 * 16 + 16; }
static int ell_win_cap(const HostLayout& H) { return (H.lds_doubles + 1) / 2 * 2; }

static thread_local int t_probe_n = 0;   // ehyb_debug_ell_stamps_probe: the stamped launch stages every window entry from three vectors
static EllArgs ell_args(ehyb_plan* P, const double* x, double* y, unsigned long long* stamps, double* xy_out = nullptr)
{
    EllArgs A;
    A.items = (const int4*)P->d_items;
    A.segs = (const int4*)P->d_segs;
    A.halo_cols = P->d_halo_cols;
    A.slab_meta = (const uint4*)P->d_slab_meta;
    A.lane_group = P->d_lane_group;
    A.slab_lrow = P->d_slab_lrow;
    A.ell_val = (const double2*)P->d_ell_val;
    A.ell_col = P->d_ell_col;
    A.x = x;
    A.y = y;
    A.win_cap = ell_win_cap(P->host);
    A.stamps = stamps;
    A.xy_out = xy_out;
    A.reverse = 0;
    A.reverse_items = 0;
    A.probe_n = stamps ? t_probe_n : 0;
    // a stream the 256 MB Infinity Cache holds whole stays there from one multiply to the next: plain loads (120 k rows, 65 MB: 15.4 us, with the hint
    // 16.2); cfg.ell_nt = 1 forces the hint.  (launch_ell_impl lowers nt_slabs for an alternating walk with cfg.ell_nt = 3)
    A.nt_slabs = (P->cfg.ell_nt == 2 || (P->cfg.ell_nt != 1 && P->host.stats.bytes_format_ell <= (256ll << 20))) ? 0 : 1024;
    // on by default: plain storage 143 -> 134 us on the audikw_1-like matrix; cfg.xcd_map = 2 for the A/B
    A.xcd_map = P->host.sym ? 0 : (P->cfg.xcd_map != 2 ? 1 : 0);
    A.item_map = P->d_item_map;  // symmetric pairs: items are sorted heaviest first, dispatched in that order
    A.windowless_zero = P->host.pb_assign ? 0 : 1;
    return A;
}

// ell_variant: 0/1 = LDS slab counter (default), 3 = static round-robin (A/B arm, tools/sweep.py --variants)

// The walk direction a caller asked for (ehyb_spmv_walk) while its call is on the stack: -1 = the plan's own alternation.
static thread_local int t_walk = -1;
namespace {
struct WalkScope {
    int saved;
    explicit WalkScope(int w) : saved(t_walk) { t_walk = w; }
    ~WalkScope() { t_walk = saved; }
};
}  // namespace

template <bool STAMP>
static int launch_ell_impl(ehyb_plan* P, const double* x, double* y, hipStream_t st, bool inl, unsigned long long* stamps, double* xy_out = nullptr)
{
    const HostLayout& H = P->host;
    const int n_items = (int)(H.items.size() / 8);
    if (n_items == 0 || H.direct) return EHYB_OK;  // direct shape: the row-segment kernel does everything
    if (H.pb_assign && H.segs.empty()) return EHYB_OK;  // no partition kept its window: pass 2 of the panel residual assigns every row
    const size_t lds = ell_lds_bytes(H);
    const bool dyn = P->cfg.ell_variant != 3;
    EllArgs A = ell_args(P, x, y, stamps, xy_out);
    // (automatic: where the stream does not fit the cache but the cache is still a fair share of it -- the walk from the short slabs
    // up costs the tail of a workgroup a few per cent: audikw_1-like, 439 MB, 83.5 -> 76.0 us; every entry stored, 729 MB, 143.3 ->
    // 136.8; 120 k rows, 65 MB, 15.4 -> 16.2; kkt3d-200, 2.56 GB, 501.9 -> 473.7 once the items are taken from the far end too)
    const bool alternates = P->cfg.ell_alternate == 1 || (P->cfg.ell_alternate == 0 && H.stats.bytes_format_ell > (256ll << 20) && H.stats.bytes_format_ell <= (8192ll << 20));
    if (!STAMP && (t_walk >= 0 || alternates)) {
        // the caller's explicit direction (ehyb_spmv_walk), else the plan's own alternation: an atomic flip, so that every one of
        // several threads launching the same plan draws a direction (plain storage: the result does not depend on it)
        A.reverse = t_walk >= 0 ? (t_walk & 1) : (P->launch_parity.fetch_xor(1, std::memory_order_relaxed) & 1);
        // more than one round of workgroups: what ran in the last round is what the cache holds, so it runs first now
        A.reverse_items = (A.reverse && n_items > kNumCU * (lds > 80 * 1024 ? 1 : 2)) ? 1 : 0;
        // cfg.ell_nt = 3: the END of every walk -- the share of the stream the 256 MB Infinity Cache can hold -- is read with plain loads,
        // so that it is still there when the next launch starts from that end
        // (half, three quarters and five quarters of that share measured level: profiles/r04_nt_hints_ab.txt)
        if (P->cfg.ell_nt == 3 && !STAMP) {
            const double keep = std::min(1.0, (double)(256ll << 20) / (double)std::max<long long>(1, H.stats.bytes_format_ell));
            A.nt_slabs = (int)(1024.0 * (1.0 - keep));
        }
    }
    const bool sym = H.sym;
#define ELL_GO(T, M, I, S)                                                                                  \
    {                                                                                                        \
        if (STAMP) HIP_TRY(hipFuncSetAttribute((const void*)ehyb_ell_kernel<T, M, STAMP, I, S>,              \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));       \
        hipLaunchKernelGGL((ehyb_ell_kernel<T, M, STAMP, I, S>), dim3(n_items), dim3(T), lds, st, A);        \
    }
#define ELL_MODE(T, I, S)           \
    if (dyn) ELL_GO(T, true, I, S)  \
    else ELL_GO(T, false, I, S)
#define ELL_INL(T, S)                 \
    if (inl) { ELL_MODE(T, true, S) } \
    else { ELL_MODE(T, false, S) }
#define ELL_LAUNCH(T)           \
    if (sym) { ELL_INL(T, true) } \
    else { ELL_INL(T, false) }
    switch (P->cfg.threads) {
        case 256: ELL_LAUNCH(256) break;
        case 512: ELL_LAUNCH(512) break;
        case 1024: ELL_LAUNCH(1024) break;
        default: EHYB_FAIL(EHYB_ERR_ARG, "ELL workgroup size %d not built (256/512/1024)", P->cfg.threads);
    }
#undef ELL_LAUNCH
#undef ELL_INL
#undef ELL_MODE
#undef ELL_GO
    HIP_TRY(hipGetLastError());
    return EHYB_OK;
}

static int launch_ell(ehyb_plan* P, const double* x, double* y, hipStream_t st, bool inl)
{
    return launch_ell_impl<false>(P, x, y, st, inl, nullptr);
}

// y = A x with x . y as a by-product (ehyb_cg.hip): possible where ONE ELL launch writes the final y of every row from a
// window that holds the row's own x -- no residual launch of its own (empty or inline residual), halo windows, not the direct
// shape.  -> number of partial sums the launch leaves (one per workgroup), 0 = not this plan.
int ehyb::spmv_xy_partials(const ehyb_plan* P)
{
    const HostLayout& H = P->host;
    const int n_items = (int)(H.items.size() / 8);
    if (!P->uploaded || H.direct || n_items == 0 || H.pb_assign || P->cfg.window_mode != EHYB_WINDOW_HALO) return 0;
    if (H.stats.nnz_er > 0 && !H.inline_er) return 0;
    return n_items;
}
int ehyb::spmv_xy(ehyb_plan* P, const double* x, double* y, void* stream, double* xy_partials)
{
    if (spmv_xy_partials(P) == 0 || !x || !y || !xy_partials) EHYB_FAIL(EHYB_ERR_STATE, "spmv_xy: not a plan that multiplies in one window launch");
    return launch_ell_impl<false>(P, x, y, (hipStream_t)stream, P->host.inline_er, nullptr, xy_partials);
}

// which: 1 = pass 1 (scale), 2 = pass 2 (reduce), 3 = both; pass 1 over the items [unit_begin, unit_end) (-1: all)
// first pass-2 unit whose rows lie at or behind cfg.row_split (= number of units: no split)
static int pass2_split(const ehyb_plan* P)
{
    const HostLayout& H = P->host;
    const int u2 = (int)(H.pb_units2.size() / 4);
    if (P->cfg.row_split <= 0) return u2;
    int lo = 0, hi = u2;   // units ascend by first row
    while (lo < hi) {
        const int mid = (lo + hi) / 2;
        if (H.pb_units2[(size_t)mid * 4 + 2] < P->cfg.row_split) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

// u2_part (pass 2): 0 = every row block, 1 = the blocks in front of cfg.row_split, 2 = the blocks from it on
static int launch_panel(ehyb_plan* P, const double* x, double* y, hipStream_t st, int probe, int which, int unit_begin = 0, int unit_end = -1, int u2_part = 0)
{
    const HostLayout& H = P->host;
    const int u_all = (int)(H.pb_items1.size() / 2), u2_all = (int)(H.pb_units2.size() / 4);
    const int u2_cut = u2_part ? pass2_split(P) : 0;
    const int u2_first = u2_part == 2 ? u2_cut : 0, u2 = (u2_part == 1 ? u2_cut : u2_all) - u2_first;
    if (unit_end < 0) unit_end = u_all;
    if (unit_begin < 0 || unit_end > u_all || unit_begin > unit_end) EHYB_FAIL(EHYB_ERR_ARG, "launch_panel: items [%d, %d) of %d", unit_begin, unit_end, u_all);
    const int u1 = unit_end - unit_begin;
    if ((which & 1) && u1 > 0) {
        // panels of up to 9,728 columns: two 512-thread workgroups per CU (one stages while the other streams); wider
        // panels leave room for one workgroup only, which then gets the CU's 16 waves
        const bool wide = P->cfg.er_panel_threads ? P->cfg.er_panel_threads == 1024 : H.pb_panel_cols > 9728;
        const bool dpp = P->cfg.er_sums != 2;
        const int xcd = P->cfg.xcd_map != 2 ? 1 : 0;
        // cfg.er_queue: one resident round of workgroups taking items from per-XCD queues (with stealing) instead of one
        // workgroup per item; needs the XCD map's contiguous eighths, and more items than workgroups to be worth it
        const int resident = kNumCU * (wide ? 1 : 2);
        // (automatic, cfg.er_queue = 0: from six items per resident workgroup up -- a workgroup that takes neighbouring items finds the panel
        // of the previous one still staged, and the XCDs even out: R-MAT 2^24 574 -> 544 us; with three or four items per workgroup the two
        // barriers and the atomic round trip per item cost more than that: 2^22 132 -> 138 us.  profiles/r04_d_panel_two_ab.jsonl)
        const bool want_queue = P->cfg.er_queue == 1 || (P->cfg.er_queue == 0 && u1 >= 6 * resident);
        int* queue = (want_queue && xcd && u1 > resident) ? P->d_pb_queue : nullptr;
        const int grid = queue ? resident : u1;
        // successive launches walk the entry stream in alternating directions (cfg.ell_alternate) where it does not fit the cache
        int rev = 0;
        if (!probe && (t_walk >= 0 || P->cfg.ell_alternate == 1 || (P->cfg.ell_alternate == 0 && H.pb_bytes > (256ll << 20)))) {
            // one direction per MULTIPLY: a multiply in parts (ehyb_spmv_part: one pass-1 launch per column segment) turns around
            // with its first part
            if (t_walk >= 0) rev = t_walk & 1;
            else if (unit_begin == 0) rev = (P->panel_parity.fetch_xor(1, std::memory_order_relaxed) ^ 1) & 1;
            else rev = P->panel_parity.load(std::memory_order_relaxed) & 1;
        }
#define PB_SCALE_P(T, D, PR)                                                                                                    \
    hipLaunchKernelGGL((ehyb_pb_scale_kernel<T, D, PR>), dim3(grid), dim3(T), (size_t)(H.pb_panel_cols + ((D) ? 0 : (T)) + 1) * 8, st, (const int2*)P->d_pb_items1 + unit_begin, (const int4*)P->d_pb_units1, \
                       P->d_pb_val, P->d_pb_colf, P->d_pb_chunk, P->d_pb_jump, x, P->d_pb_partial, H.pb_panel_cols, probe, xcd, queue, u1, rev)
#define PB_SCALE(T, D)                  \
    if (probe) PB_SCALE_P(T, D, true);  \
    else PB_SCALE_P(T, D, false)
        if (wide) {
            if (dpp) { PB_SCALE(1024, true); } else { PB_SCALE(1024, false); }
        } else {
            if (dpp) { PB_SCALE(512, true); } else { PB_SCALE(512, false); }
        }
#undef PB_SCALE
#undef PB_SCALE_P
    }
    if ((which & 2) && u2 > 0) {
        // (cfg.er_nt: 0 = by the size of what pass 2 reads -- 10 B per partial sum -- against half the 256 MB Infinity Cache, 1 / 2 = always / never)
        const bool nt = P->cfg.er_nt == 1 || (P->cfg.er_nt == 0 && H.pb_partials * 10 > (128ll << 20));
        if (nt)
            hipLaunchKernelGGL((ehyb_pb_reduce_kernel<512, true>), dim3(u2), dim3(512), (size_t)H.pb_rows_max * 8, st, (const int4*)P->d_pb_units2 + u2_first,
                               P->d_pb_partial, P->d_pb_row, y, probe);
        else
            hipLaunchKernelGGL((ehyb_pb_reduce_kernel<512, false>), dim3(u2), dim3(512), (size_t)H.pb_rows_max * 8, st, (const int4*)P->d_pb_units2 + u2_first,
                               P->d_pb_partial, P->d_pb_row, y, probe);
    }
    HIP_TRY(hipGetLastError());
    return EHYB_OK;
}

static int launch_er(ehyb_plan* P, const double* x, double* y, hipStream_t st)
{
    const HostLayout& H = P->host;
    if (H.er_bins[3] == 0 && !H.er_panel) return EHYB_OK;  // (er_bins[3] = CSR segments: a device-built panel form has none)
    if (H.er_panel) {  // panel form: scale (x panels in LDS) then reduce (y blocks in LDS)
        return launch_panel(P, x, y, st, 0, 3);  // (probe arms only through ehyb_debug_panel_times)
    }
    const int n_blocks = (int)(H.er_blocks.size() / 4);
    if (P->cfg.er_threads != 256) EHYB_FAIL(EHYB_ERR_ARG, "residual workgroup size %d not built (256)", P->cfg.er_threads);
    if (H.direct)
        hipLaunchKernelGGL((ehyb_er_kernel<256, true>), dim3(n_blocks), dim3(256), 0, st, (const int4*)P->d_er_blocks,
                           P->d_er_seg_ptr, P->d_er_seg_row, P->d_er_col, P->d_er_val, x, y);
    else
        hipLaunchKernelGGL((ehyb_er_kernel<256, false>), dim3(n_blocks), dim3(256), 0, st, (const int4*)P->d_er_blocks,
                           P->d_er_seg_ptr, P->d_er_seg_row, P->d_er_col, P->d_er_val, x, y);
    HIP_TRY(hipGetLastError());
    return EHYB_OK;
}

// Where the residual runs (decided by the layout builder, HostLayout::inline_er).  Its own launch
// costs a second ~8 us kernel boundary but gives the residual thousands of independent blocks;
// inline -- every ELL lane adds its row's few residual entries before writing y -- costs nothing
// when the residual is tiny and would serialise a divergent per-lane loop when it is not.
// fuse_er: 1 = always inline, 2 = never, 0 = automatic: inline iff the residual holds < 0.2 % of
// the entries.  Multi-GPU plans keep the phases apart (phase 1 reads only the rank's x segment).
static bool fuse_residual(const ehyb_plan* P) { return P->host.inline_er; }

template <class T, class A>
static int upload(T** dst, const std::vector<T, A>& src)
{
    *dst = nullptr;
    size_t bytes = std::max<size_t>(src.size(), 1) * sizeof(T) + 4096;  // slack: clamped prefetches
    HIP_TRY(hipMalloc((void**)dst, bytes));
    if (!src.empty()) HIP_TRY(hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
    return EHYB_OK;
}

static void free_device(ehyb_plan* P)
{
    void** ptrs[] = {(void**)&P->d_halo_cols,  (void**)&P->d_ell_val,   (void**)&P->d_ell_col,    (void**)&P->d_lane_group,
                     (void**)&P->d_slab_meta,  (void**)&P->d_items,     (void**)&P->d_segs,       (void**)&P->d_er_seg_ptr,
                     (void**)&P->d_er_seg_row, (void**)&P->d_er_col,    (void**)&P->d_er_val,     (void**)&P->d_er_blocks,
                     (void**)&P->d_slab_lrow,  (void**)&P->d_pb_val,    (void**)&P->d_pb_colf,    (void**)&P->d_pb_chunk,   (void**)&P->d_pb_jump,
                     (void**)&P->d_pb_units1,  (void**)&P->d_pb_items1,  (void**)&P->d_pb_queue,   (void**)&P->d_pb_row,    (void**)&P->d_pb_units2,  (void**)&P->d_pb_partial,
                     (void**)&P->d_item_map,   (void**)&P->d_ell_src,    (void**)&P->d_ell_src2,  (void**)&P->d_er_src,     (void**)&P->d_pb_src};
    for (void** q : ptrs) {
        if (*q) (void)hipFree(*q);
        *q = nullptr;
    }
    for (int32_t* q : P->retired_item_maps) (void)hipFree(q);
    P->retired_item_maps.clear();
    P->item_map.clear();
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
int ehyb_stream_create(void** stream)
{
    if (!stream) EHYB_FAIL(EHYB_ERR_ARG, "ehyb_stream_create: null");
    hipStream_t s = nullptr;
    HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *stream = (void*)s;
    return EHYB_OK;
}
int ehyb_stream_destroy(void* stream)
{
    if (stream) HIP_TRY(hipStreamDestroy((hipStream_t)stream));
    return EHYB_OK;
}
int ehyb_stream_sync(void* stream)
{
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return EHYB_OK;
}
int ehyb_dev_mem_info(size_t* free_bytes, size_t* total_bytes)
{
    if (!free_bytes || !total_bytes) EHYB_FAIL(EHYB_ERR_ARG, "ehyb_dev_mem_info: null argument");
    HIP_TRY(hipMemGetInfo(free_bytes, total_bytes));
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

// Diagnostic (tools/stamps.py): one launch of the stamped instantiation of the ELL kernel.
// out[4*i + {0,1,2,3}] = entry / staged / exit wall-clock ticks (100 MHz) and XCC id of item i.
int ehyb_debug_ell_stamps(ehyb_plan* P, const double* x, double* y, unsigned long long* out_host);
int ehyb_debug_ell_stamps_probe(ehyb_plan* P, const double* x, double* y, unsigned long long* out_host, int triple_gather)
{
    t_probe_n = (triple_gather == 2) ? -1 : (triple_gather && P) ? P->host.n_cols : 0;   // 2: no halo gather (diagnostic)
    const int rc = ehyb_debug_ell_stamps(P, x, y, out_host);
    t_probe_n = 0;
    return rc;
}

int ehyb_debug_ell_stamps(ehyb_plan* P, const double* x, double* y, unsigned long long* out_host)
{
    if (!P || !P->uploaded || !out_host) EHYB_FAIL(EHYB_ERR_ARG, "ehyb_debug_ell_stamps: bad arguments");
    const int n_items = (int)(P->host.items.size() / 8);
    unsigned long long* d = nullptr;
    HIP_TRY(hipMalloc((void**)&d, (size_t)n_items * 32));
    HIP_TRY(hipMemset(d, 0, (size_t)n_items * 32));
    int rc = launch_ell_impl<true>(P, x, y, nullptr, P->host.inline_er, d);
    if (rc == EHYB_OK && hipDeviceSynchronize() != hipSuccess) rc = EHYB_ERR_HIP;
    if (rc == EHYB_OK && hipMemcpy(out_host, d, (size_t)n_items * 32, hipMemcpyDeviceToHost) != hipSuccess) rc = EHYB_ERR_HIP;
    if (rc == EHYB_ERR_HIP) set_error("ehyb_debug_ell_stamps: %s", hipGetErrorString(hipGetLastError()));
    (void)hipFree(d);
    return rc;
}

int ehyb_halo_step(ehyb_plan* P, const double* x, double* y, const int32_t* send_idx, double* send_buf, int64_t n_send, int n_chunks,
                   ehyb_exchange_fn exchange, void* user, void* compute_stream, void* comm_stream)
{
    if (!P || !exchange || n_chunks < 1) EHYB_FAIL(EHYB_ERR_ARG, "ehyb_halo_step: bad arguments");
    int n_segs = 1;
    (void)ehyb_plan_col_segs(P, &n_segs);
    if (n_segs != n_chunks + 1) EHYB_FAIL(EHYB_ERR_ARG, "ehyb_halo_step: the plan has %d column segments, %d chunks need %d", n_segs, n_chunks, n_chunks + 1);
    int rc = ehyb_step_pack(x, send_idx, send_buf, n_send, compute_stream, comm_stream);
    if (rc == EHYB_OK) rc = ehyb_step_part(P, x, y, compute_stream, comm_stream, 0, 0, 1, EHYB_PART_FIRST);
    for (int k = 0; k < n_chunks && rc == EHYB_OK; ++k) {
        if (exchange(k, comm_stream, user) != 0) EHYB_FAIL(EHYB_ERR_STATE, "ehyb_halo_step: the caller's exchange of chunk %d failed", k);
        rc = ehyb_step_part(P, x, y, compute_stream, comm_stream, 1, 1 + k, 2 + k, k == n_chunks - 1 ? EHYB_PART_LAST : 0);
    }
    return rc;
}

// Tuning of the item -> workgroup map on the device the plan lives on (see ehyb.h).
int ehyb_plan_tune(ehyb_plan* P, const double* x, double* y, int reps, double* span_before_us, double* span_after_us)
{
    clear_error();
    if (!P || !P->uploaded || !x || !y) EHYB_FAIL(EHYB_ERR_ARG, "ehyb_plan_tune: bad arguments");
    if (span_before_us) *span_before_us = 0;
    if (span_after_us) *span_after_us = 0;
    const HostLayout& H = P->host;
    const int n_items = (int)(H.items.size() / 8);
    const int resident = kNumCU * std::max(1, P->cfg.items_per_cu);
    // one resident round only: with more items than slots the later ones start wherever a CU falls free
    if (H.direct || n_items < 16 || n_items > resident || (H.pb_assign && H.segs.empty())) return EHYB_OK;
    reps = std::min(std::max(reps, 1), 16);
    unsigned long long* d = nullptr;
    HIP_TRY(hipMalloc((void**)&d, (size_t)n_items * 32));
    std::vector<unsigned long long> st((size_t)n_items * 4);
    // cost of an item: the bytes its slabs stream (values 16 B per lane and pair + column words) + its windows
    std::vector<double> cost((size_t)n_items, 0.0);
    for (int i = 0; i < n_items; ++i) {
        const int32_t* rec = &H.items[(size_t)i * 8];
        double c = 0;
        for (int s = rec[2]; s < rec[3]; ++s) {
            const uint32_t w = H.slab_meta[(size_t)s * 4 + 3];
            c += (double)(w >> 16) * (1024.0 + 4.0 * ((w & 0x3Fu) + 1));
        }
        for (int g = rec[0]; g < rec[1]; ++g) c += 8.0 * (H.segs[(size_t)g * 8 + 6] + H.segs[(size_t)g * 8 + 3]) + 16.0 * (H.segs[(size_t)g * 8 + 5] - H.segs[(size_t)g * 8 + 4]);
        cost[(size_t)i] = c + 1.0;
    }
    auto measure = [&](std::vector<double>* dur, std::vector<int>* xcc, double* span_us) -> int {
        // item taken by block b under the CURRENT map
        dur->assign((size_t)n_items, 0.0);
        xcc->assign((size_t)n_items, -1);
        double span = 0;
        for (int r = 0; r < reps + 1; ++r) {  // the first launch warms the caches and is not counted
            if (hipMemset(d, 0, (size_t)n_items * 32) != hipSuccess) return EHYB_ERR_HIP;
            const int rc = launch_ell_impl<true>(P, x, y, nullptr, H.inline_er, d);
            if (rc != EHYB_OK) return rc;
            if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(st.data(), d, (size_t)n_items * 32, hipMemcpyDeviceToHost) != hipSuccess) return EHYB_ERR_HIP;
            if (r == 0) continue;
            unsigned long long t0 = ~0ull, t1 = 0;
            for (int b = 0; b < n_items; ++b) {
                (*dur)[(size_t)b] += (double)(st[(size_t)b * 4 + 2] - st[(size_t)b * 4 + 0]) / (100.0 * reps);  // 100 MHz clock -> us
                const int k = (int)st[(size_t)b * 4 + 3] & 15;
                if ((*xcc)[(size_t)b] >= 0 && (*xcc)[(size_t)b] != k) (*xcc)[(size_t)b] = -2;  // not a stable placement
                else if ((*xcc)[(size_t)b] != -2) (*xcc)[(size_t)b] = k;
                t0 = std::min(t0, st[(size_t)b * 4 + 0]);
                t1 = std::max(t1, st[(size_t)b * 4 + 2]);
            }
            span += (double)(t1 - t0) / (100.0 * reps);
        }
        *span_us = span;
        return EHYB_OK;
    };
    std::vector<double> dur;
    std::vector<int> xcc;
    double span0 = 0, span1 = 0;
    int rc = measure(&dur, &xcc, &span0);
    std::vector<int32_t> map_now((size_t)n_items);
    for (int b = 0; b < n_items; ++b) map_now[(size_t)b] = P->item_map.empty() ? (P->host.sym || P->cfg.xcd_map == 2 ? b : [&] { const int k = b & 7, j = b >> 3, chunk = n_items >> 3, rem = n_items & 7; return k * chunk + std::min(k, rem) + j; }()) : P->item_map[(size_t)b];
    bool stable = rc == EHYB_OK;
    for (int b = 0; b < n_items && stable; ++b) stable = xcc[(size_t)b] >= 0;
    if (stable) {
        // how fast each XCD streamed what it was given: median over its workgroups of bytes per microsecond
        std::vector<std::vector<double>> rates(16);
        std::vector<std::vector<int>> slots(16);
        for (int b = 0; b < n_items; ++b) {
            rates[(size_t)xcc[(size_t)b]].push_back(cost[(size_t)map_now[(size_t)b]] / std::max(dur[(size_t)b], 1e-3));
            slots[(size_t)xcc[(size_t)b]].push_back(b);
        }
        std::vector<std::pair<double, int>> order;  // (median rate, xcd), fastest first
        for (int k = 0; k < 16; ++k)
            if (!rates[(size_t)k].empty()) {
                std::nth_element(rates[(size_t)k].begin(), rates[(size_t)k].begin() + rates[(size_t)k].size() / 2, rates[(size_t)k].end());
                order.push_back({-rates[(size_t)k][rates[(size_t)k].size() / 2], k});
            }
        std::sort(order.begin(), order.end());
        std::vector<int> by_cost((size_t)n_items);
        for (int i = 0; i < n_items; ++i) by_cost[(size_t)i] = i;
        std::stable_sort(by_cost.begin(), by_cost.end(), [&](int a, int b2) { return cost[(size_t)a] > cost[(size_t)b2]; });
        // the heaviest items on the fastest XCD, and so on down: the pairing that minimises the largest cost / rate
        std::vector<int32_t> map_new((size_t)n_items, -1);
        size_t at = 0;
        for (const auto& o : order)
            for (int b : slots[(size_t)o.second]) map_new[(size_t)b] = by_cost[at++];
        int32_t* dm = nullptr;
        if (at == (size_t)n_items && hipMalloc((void**)&dm, (size_t)n_items * 4) == hipSuccess &&
            hipMemcpy(dm, map_new.data(), (size_t)n_items * 4, hipMemcpyHostToDevice) == hipSuccess) {
            int32_t* old = P->d_item_map;
            std::vector<int32_t> old_host = P->item_map;
            P->d_item_map = dm;
            P->item_map = map_new;
            std::vector<double> dur2;
            std::vector<int> xcc2;
            rc = measure(&dur2, &xcc2, &span1);
            if (rc != EHYB_OK || span1 >= span0) {  // no gain on this device: the map the plan had stays
                P->d_item_map = old;
                P->item_map = old_host;
                (void)hipFree(dm);
                span1 = span0;
            } else if (old) {
                P->retired_item_maps.push_back(old);  // a hipGraph captured earlier may still name it: freed with the plan
            }
        } else if (dm) {
            (void)hipFree(dm);
        }
    }
    (void)hipFree(d);
    if (span_before_us) *span_before_us = span0;
    if (span_after_us) *span_after_us = stable ? span1 : span0;
    return rc;
}

// Diagnostic (tools/panel_sweep.py): mean time of each pass of the panel residual alone, `iters` launches each
// between HIP events on the null stream; probe switches single steps of the kernels off (results wrong then).
int ehyb_debug_panel_times(ehyb_plan* P, const double* x, double* y, int iters, int probe, double* ms_scale, double* ms_reduce)
{
    if (!P || !P->uploaded || !P->host.er_panel || iters < 1 || !ms_scale || !ms_reduce) EHYB_FAIL(EHYB_ERR_ARG, "ehyb_debug_panel_times: bad arguments");
    hipEvent_t a, b;
    HIP_TRY(hipEventCreate(&a));
    HIP_TRY(hipEventCreate(&b));
    double* out[2] = {ms_scale, ms_reduce};
    int rc = EHYB_OK;
    for (int which = 1; which <= 2 && rc == EHYB_OK; ++which) {
        for (int i = 0; i < 3 && rc == EHYB_OK; ++i) rc = launch_panel(P, x, y, nullptr, probe, which);
        (void)hipEventRecord(a, nullptr);
        for (int i = 0; i < iters && rc == EHYB_OK; ++i) rc = launch_panel(P, x, y, nullptr, probe, which);
        (void)hipEventRecord(b, nullptr);
        (void)hipEventSynchronize(b);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, a, b);
        *out[which - 1] = ms / iters;
    }
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    return rc;
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
    if (H.deferred.pending) EHYB_FAIL(EHYB_ERR_STATE, "ehyb_plan_upload: the plan's panel form was left to the device and never built");
    int rc;
#define UP(dst, src)                           \
    if ((rc = upload(&P->dst, H.src)) != EHYB_OK) { \
        free_device(P);                        \
        return rc;                             \
    }
    UP(d_halo_cols, halo_cols)
    UP(d_ell_val, ell_val)
    UP(d_ell_col, ell_col)
    UP(d_lane_group, lane_group)
    UP(d_slab_meta, slab_meta)
    UP(d_items, items)
    UP(d_segs, segs)
    UP(d_slab_lrow, slab_lrow)
    if (H.er_panel) {
        // the residual launch runs the panel form: the CSR segments stay on the host
        if (!H.pb_host_missing) {  // (else: built where they are, er_panel_dev.hip)
            UP(d_pb_val, pb_val)
            UP(d_pb_colf, pb_colf)
            UP(d_pb_chunk, pb_chunk)
            UP(d_pb_jump, pb_jump)
            UP(d_pb_row, pb_row)
        }
        UP(d_pb_units1, pb_units1)
        UP(d_pb_items1, pb_items1)
        UP(d_pb_units2, pb_units2)
        if (hipMalloc((void**)&P->d_pb_queue, 256 * sizeof(int)) != hipSuccess || hipMemset(P->d_pb_queue, 0, 256 * sizeof(int)) != hipSuccess) {
            free_device(P);
            EHYB_FAIL(EHYB_ERR_HIP, "ehyb_plan_upload: no device memory for the work queues");
        }
        if (hipMalloc((void**)&P->d_pb_partial, (size_t)std::max<int64_t>(H.pb_partials, 1) * 8) != hipSuccess) {
            free_device(P);
            EHYB_FAIL(EHYB_ERR_HIP, "ehyb_plan_upload: no device memory for %lld partial sums", (long long)H.pb_partials);
        }
    } else {
        UP(d_er_seg_ptr, er_seg_ptr)
        UP(d_er_seg_row, er_seg_row)
        UP(d_er_col, er_col)
        UP(d_er_val, er_val)
        UP(d_er_blocks, er_blocks)
    }
#undef UP
    // opt in to the full 160 KiB of LDS (the role of cudaFuncSetAttribute at kernel.cu:351,411).  The
    // attribute belongs to the kernel, not to a plan: it is set to the device maximum, so plans with
    // windows of different sizes can live side by side in one process.
    const int lds = EHYB_LDS_MAX_DOUBLES * 8;
#define LDS_ATTR(K) HIP_TRY(hipFuncSetAttribute((const void*)(K), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
#define LDS_ATTR_S(T, S)                                   \
    LDS_ATTR((ehyb_ell_kernel<T, false, false, false, S>)) \
    LDS_ATTR((ehyb_ell_kernel<T, false, false, true, S>))  \
    LDS_ATTR((ehyb_ell_kernel<T, true, false, false, S>))  \
    LDS_ATTR((ehyb_ell_kernel<T, true, false, true, S>))
#define LDS_ATTR_T(T)      \
    LDS_ATTR_S(T, false)   \
    LDS_ATTR_S(T, true)
    LDS_ATTR_T(256)
    LDS_ATTR_T(512)
    LDS_ATTR_T(1024)
    LDS_ATTR((ehyb_pb_scale_kernel<512, true, false>))
    LDS_ATTR((ehyb_pb_scale_kernel<512, false, false>))
    LDS_ATTR((ehyb_pb_scale_kernel<1024, true, false>))
    LDS_ATTR((ehyb_pb_scale_kernel<1024, false, false>))
    LDS_ATTR((ehyb_pb_scale_kernel<512, true, true>))
    LDS_ATTR((ehyb_pb_scale_kernel<512, false, true>))
    LDS_ATTR((ehyb_pb_scale_kernel<1024, true, true>))
    LDS_ATTR((ehyb_pb_scale_kernel<1024, false, true>))
    LDS_ATTR((ehyb_pb_reduce_kernel<512, true>))
    LDS_ATTR((ehyb_pb_reduce_kernel<512, false>))
#undef LDS_ATTR_T
#undef LDS_ATTR_S
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
    if (P->host.direct && phase != 0) EHYB_FAIL(EHYB_ERR_STATE, "ehyb_spmv_phase: a plan in the direct shape (small matrix) has no phases");
    if (phase == 0 || phase == 1) rc = launch_ell(P, x, y, st, false);
    if (rc == EHYB_OK && (phase == 0 || phase == 2)) rc = launch_er(P, x, y, st);
    return rc;
}

int ehyb_spmv(ehyb_plan* P, const double* x, double* y, void* stream)
{
    return ehyb_spmv_phase(P, x, y, stream, 0);
}

int ehyb_spmv_walk(ehyb_plan* P, const double* x, double* y, void* stream, int walk)
{
    if (walk < -1 || walk > 1) EHYB_FAIL(EHYB_ERR_ARG, "ehyb_spmv_walk: walk %d (EHYB_WALK_AUTO, _FIRST_TO_LAST, _LAST_TO_FIRST)", walk);
    WalkScope w(walk);
    return ehyb_spmv_phase(P, x, y, stream, 0);
}

// ---- a captured multiply (or run of multiplies) that keeps the alternation: see ehyb.h
struct ehyb_graph {
    hipGraphExec_t exec[2] = {nullptr, nullptr};   // [d]: the run starting with direction d; equal runs (even count, or a plan that does not alternate) share exec[0]
    int next = 0;
    bool two = false;
};

int ehyb_spmv_graph_create(ehyb_plan* P, const double* x, double* y, int multiplies, ehyb_graph** out)
{
    clear_error();
    if (!P || !x || !y || !out || multiplies < 1) EHYB_FAIL(EHYB_ERR_ARG, "ehyb_spmv_graph_create: bad arguments");
    *out = nullptr;
    if (!P->uploaded) EHYB_FAIL(EHYB_ERR_STATE, "ehyb_spmv_graph_create: plan not uploaded");
    hipStream_t own = nullptr;
    HIP_TRY(hipStreamCreateWithFlags(&own, hipStreamNonBlocking));
    ehyb_graph* G = new ehyb_graph;
    G->two = (multiplies & 1) != 0;   // an odd run ends on the direction it began with: the next launch must begin with the other
    int rc = EHYB_OK;
    for (int d = 0; d < (G->two ? 2 : 1) && rc == EHYB_OK; ++d) {
        hipGraph_t g = nullptr;
        if (hipStreamBeginCapture(own, hipStreamCaptureModeThreadLocal) != hipSuccess) {
            rc = EHYB_ERR_HIP;
            break;
        }
        for (int i = 0; i < multiplies && rc == EHYB_OK; ++i) rc = ehyb_spmv_walk(P, x, y, (void*)own, (d + i) & 1);
        const hipError_t e = hipStreamEndCapture(own, &g);
        if (rc == EHYB_OK && (e != hipSuccess || hipGraphInstantiate(&G->exec[d], g, nullptr, nullptr, 0) != hipSuccess)) rc = EHYB_ERR_HIP;
        if (g) (void)hipGraphDestroy(g);
    }
    (void)hipStreamDestroy(own);
    if (rc != EHYB_OK) {
        if (rc == EHYB_ERR_HIP) set_error("ehyb_spmv_graph_create: capture failed: %s", hipGetErrorString(hipGetLastError()));
        ehyb_graph_destroy(G);
        return rc;
    }
    *out = G;
    return EHYB_OK;
}

int ehyb_graph_launch(ehyb_graph* G, void* stream)
{
    if (!G || !G->exec[0]) EHYB_FAIL(EHYB_ERR_ARG, "ehyb_graph_launch: null");
    HIP_TRY(hipGraphLaunch(G->exec[G->two ? G->next : 0], (hipStream_t)stream));
    if (G->two) G->next ^= 1;
    return EHYB_OK;
}

void ehyb_graph_destroy(ehyb_graph* G)
{
    if (!G) return;
    for (auto e : G->exec)
        if (e) (void)hipGraphExecDestroy(e);
    delete G;
}

int ehyb_plan_col_segs(const ehyb_plan* P, int* n_col_segs)
{
    if (!P || !n_col_segs) EHYB_FAIL(EHYB_ERR_ARG, "ehyb_plan_col_segs: null argument");
    *n_col_segs = P->host.col_seg_first.size() >= 2 ? (int)P->host.col_seg_first.size() - 1 : 1;
    return EHYB_OK;
}

// The multiply in parts (multi-GPU: x arrives column segment by column segment).
int ehyb_spmv_part(ehyb_plan* P, const double* x, double* y, void* stream, int seg_begin, int seg_end, int flags)
{
    if (!P || !x || !y) EHYB_FAIL(EHYB_ERR_ARG, "ehyb_spmv_part: null argument");
    if (!P->uploaded) EHYB_FAIL(EHYB_ERR_STATE, "ehyb_spmv_part: plan not uploaded (no CPU fallback exists)");
    const HostLayout& H = P->host;
    int n_segs = 1;
    (void)ehyb_plan_col_segs(P, &n_segs);
    if (seg_begin < 0 || seg_end > n_segs || seg_begin > seg_end) EHYB_FAIL(EHYB_ERR_ARG, "ehyb_spmv_part: column segments [%d, %d) of %d", seg_begin, seg_end, n_segs);
    if (H.direct || fuse_residual(P)) EHYB_FAIL(EHYB_ERR_STATE, "ehyb_spmv_part: this plan multiplies in one launch (direct shape or inline residual): it has no parts");
    hipStream_t st = (hipStream_t)stream;
    int rc = EHYB_OK;
    if (flags & EHYB_PART_FIRST) rc = launch_ell(P, x, y, st, false);
    if (rc != EHYB_OK || (H.er_bins[3] == 0 && !H.er_panel)) return rc;
    if (!H.er_panel) return (flags & EHYB_PART_LAST) ? launch_er(P, x, y, st) : EHYB_OK;  // CSR residual: one launch, needs all of x
    if (seg_end > seg_begin) {
        const int ub = H.pb_seg_item.empty() ? 0 : H.pb_seg_item[(size_t)seg_begin];
        const int ue = H.pb_seg_item.empty() ? -1 : H.pb_seg_item[(size_t)seg_end];
        rc = launch_panel(P, x, y, st, 0, 1, ub, ue);
    }
    // the closing pass: all row blocks, or -- with cfg.row_split -- the foreign rows first (EHYB_PART_LAST_FOREIGN, as soon as
    // segment 0's pass 1 is enqueued) and the rows in front of the split at the end
    const bool split = P->cfg.row_split > 0;
    if (rc == EHYB_OK && (flags & EHYB_PART_LAST_FOREIGN)) {
        if (!split) EHYB_FAIL(EHYB_ERR_STATE, "ehyb_spmv_part: EHYB_PART_LAST_FOREIGN needs a plan built with cfg.row_split");
        rc = launch_panel(P, x, y, st, 0, 2, 0, -1, 2);
    }
    if (rc == EHYB_OK && (flags & EHYB_PART_LAST)) rc = launch_panel(P, x, y, st, 0, 2, 0, -1, split ? 1 : 0);
    return rc;
}

int ehyb_gather(const double* src, const int32_t* idx, double* dst, int64_t n, void* stream)
{
    if (n < 0 || (n > 0 && (!src || !idx || !dst))) EHYB_FAIL(EHYB_ERR_ARG, "ehyb_gather: bad arguments");
    if (n == 0) return EHYB_OK;
    hipLaunchKernelGGL(ehyb_gather_kernel, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, (hipStream_t)stream, src, idx, dst, (long long)n);
    HIP_TRY(hipGetLastError());
    return EHYB_OK;
}

int ehyb_scatter_add(double* y, const int32_t* idx, const double* src, int64_t n, void* stream)
{
    if (n < 0 || (n > 0 && (!y || !idx || !src))) EHYB_FAIL(EHYB_ERR_ARG, "ehyb_scatter_add: bad arguments");
    if (n == 0) return EHYB_OK;
    hipLaunchKernelGGL(ehyb_scatter_add_kernel, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, (hipStream_t)stream, y, idx, src, (long long)n);
    HIP_TRY(hipGetLastError());
    return EHYB_OK;
}

// "`waiter` waits for everything enqueued on `on` so far": an event record + a stream wait.  The events are a small
// per-thread, per-DEVICE ring, made once and reused (recording an event again while an earlier wait on it is pending is
// well defined: a wait refers to the record that preceded it); a thread that drives plans on several devices gets one ring
// for each, and the rings are destroyed when the thread ends.
namespace {
struct EventRings {
    static constexpr int kRing = 16, kDevices = 16;
    hipEvent_t ring[kDevices][kRing] = {};
    int next[kDevices] = {};
    ~EventRings()
    {
        for (auto& dev : ring)
            for (hipEvent_t e : dev)
                if (e) (void)hipEventDestroy(e);
    }
};
}  // namespace
static int stream_wait_stream(hipStream_t waiter, hipStream_t on)
{
    static thread_local EventRings R;
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= EventRings::kDevices) EHYB_FAIL(EHYB_ERR_ARG, "stream_wait_stream: device %d", dev);
    hipEvent_t& e = R.ring[dev][R.next[dev]];
    R.next[dev] = (R.next[dev] + 1) % EventRings::kRing;
    if (!e) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(e, on));
    HIP_TRY(hipStreamWaitEvent(waiter, e, 0));
    return EHYB_OK;
}

int ehyb_step_pack(const double* x, const int32_t* idx, double* send_buf, int64_t n, void* compute_stream, void* comm_stream)
{
    int rc = ehyb_gather(x, idx, send_buf, n, compute_stream);
    if (rc == EHYB_OK && comm_stream != compute_stream) rc = stream_wait_stream((hipStream_t)comm_stream, (hipStream_t)compute_stream);
    return rc;
}

int ehyb_step_part(ehyb_plan* P, const double* x, double* y, void* compute_stream, void* comm_stream, int wait_comm, int seg_begin, int seg_end,
                   int flags)
{
    if (wait_comm && comm_stream != compute_stream) {
        const int rc = stream_wait_stream((hipStream_t)compute_stream, (hipStream_t)comm_stream);
        if (rc != EHYB_OK) return rc;
    }
    return ehyb_spmv_part(P, x, y, compute_stream, seg_begin, seg_end, flags);
}

int ehyb_spmv_bench(ehyb_plan* P, const double* x, double* y, void* stream, int warmup, int iters,
                    double* ms_total, double* ms_ell, double* ms_er)
{
    if (!P || iters < 1) EHYB_FAIL(EHYB_ERR_ARG, "ehyb_spmv_bench: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    // The loop of the reference (spmv.cu:108-116): MAXIter multiplies of the same x, back to back.
    // It is replayed from a hipGraph of kBatch multiplies (the legacy default stream cannot be
    // captured: a private blocking stream stands in for it), and the host never runs more than
    // 2 x kThrottle multiplies ahead of the device: with 15 us kernels an unthrottled loop of a few
    // hundred launches outran the device far enough to hit a one-off ~80 ms stall inside the
    // runtime, which a 500-iteration measurement reported as 183 us per multiply instead of 15.
    constexpr int kBatch = 32, kThrottle = 256;
    struct Loop {
        hipStream_t own = nullptr;
        hipGraph_t graph = nullptr;
        hipGraphExec_t exec = nullptr;
        hipEvent_t a = nullptr, b = nullptr, t[2] = {nullptr, nullptr};
        ~Loop()
        {
            if (exec) (void)hipGraphExecDestroy(exec);
            if (graph) (void)hipGraphDestroy(graph);
            for (hipEvent_t e : {a, b, t[0], t[1]})
                if (e) (void)hipEventDestroy(e);
            if (own) (void)hipStreamDestroy(own);
        }
    } L;
    if (!st) {
        HIP_TRY(hipStreamCreate(&L.own));
        st = L.own;
        stream = (void*)L.own;
    }
    int rc;
    for (int i = 0; i < warmup; ++i)
        if ((rc = ehyb_spmv(P, x, y, stream)) != EHYB_OK) return rc;
    if (P->cfg.graphs != 2 && iters >= 2 * kBatch && hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) == hipSuccess) {
        int erc = EHYB_OK;
        for (int i = 0; i < kBatch && erc == EHYB_OK; ++i) erc = ehyb_spmv(P, x, y, stream);
        const hipError_t eend = hipStreamEndCapture(st, &L.graph);
        if (erc != EHYB_OK || eend != hipSuccess || hipGraphInstantiate(&L.exec, L.graph, nullptr, nullptr, 0) != hipSuccess)
            L.exec = nullptr;
        (void)hipGetLastError();
    }
    HIP_TRY(hipEventCreate(&L.a));
    HIP_TRY(hipEventCreate(&L.b));
    HIP_TRY(hipEventCreate(&L.t[0]));
    HIP_TRY(hipEventCreate(&L.t[1]));
    HIP_TRY(hipEventRecord(L.a, st));
    int done = 0, marks = 0, since = 0;
    while (done < iters) {
        if (L.exec && done + kBatch <= iters) {
            HIP_TRY(hipGraphLaunch(L.exec, st));
            done += kBatch;
            since += kBatch;
        } else {
            if ((rc = ehyb_spmv(P, x, y, stream)) != EHYB_OK) return rc;
            ++done;
            ++since;
        }
        if (since >= kThrottle) {  // wait for the mark before the one just set
            HIP_TRY(hipEventRecord(L.t[marks & 1], st));
            if (marks > 0) HIP_TRY(hipEventSynchronize(L.t[(marks - 1) & 1]));
            ++marks;
            since = 0;
        }
    }
    HIP_TRY(hipEventRecord(L.b, st));
    HIP_TRY(hipEventSynchronize(L.b));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, L.a, L.b));
    if (ms_total) *ms_total = ms;
    if (ms_ell || ms_er) {
        const int n = std::min(iters, 200);
        struct Events {  // destroyed on every way out of this block
            std::vector<hipEvent_t> v;
            ~Events()
            {
                for (hipEvent_t e : v)
                    if (e) (void)hipEventDestroy(e);
            }
        } evs;
        evs.v.assign((size_t)3 * n, nullptr);
        std::vector<hipEvent_t>& ev = evs.v;
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
    return ehyb_plan_create_segs(m, 0, m->dimension, cfg, 0, nullptr, plan);
}

// Build + upload.  What the device can build is left to it (cfg.symbolic): the host lays out the windows and leaves a
// panel-form residual as the entries in row order; the device deals them out (er_panel_dev.hip).
int ehyb_plan_create_segs(const matrixCOO* m, int row_begin, int row_end, const ehyb_config* cfg, int n_col_segs, const int* col_seg_first,
                          ehyb_plan** plan)
{
    clear_error();
    if (!plan) EHYB_FAIL(EHYB_ERR_ARG, "ehyb_plan_create: null output");
    *plan = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count < 1)
        EHYB_FAIL(EHYB_ERR_NO_DEVICE, "ehyb_plan_create: no HIP device visible (the EHYB multiply has no CPU fallback)");
    const double t0 = wall_seconds();
    // The device builder holds ~80 bytes of temporaries per residual entry at its peak (sort keys and payloads twice, rows,
    // columns, values, the streams): where the device has not got that free -- every entry of the rows taken as residual, the
    // most it can be -- the panel form is built on the host instead (same arrays, slower).
    bool on_device_ok = true;
    if (m && m->rowIdx && row_begin >= 0 && row_end <= m->dimension && row_begin < row_end) {
        size_t free_b = 0, total_b = 0;
        const double need = 80.0 * (double)((int64_t)m->rowIdx[row_end] - m->rowIdx[row_begin]) + 268435456.0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || (double)free_b < need) on_device_ok = false;
    }
    int rc = create_host_plan(m, row_begin, row_end, cfg, n_col_segs, col_seg_first, on_device_ok, plan);
    if (rc != EHYB_OK) return rc;
    const double t1 = wall_seconds();
    bool on_device = (*plan)->host.deferred.pending;
    if (on_device) rc = build_panel_on_device(*plan);
    if (on_device && rc != EHYB_OK && (rc == EHYB_ERR_HIP || rc == EHYB_ERR_ALLOC)) {
        // the device ran out of memory after all (ranks that share a device can each pass the check above before the other
        // allocates) or a sort failed: the matrix is still with the caller -- build the whole layout again with the panel form on the
        // host (the same arrays, slower) instead of giving up
        const std::string why = ehyb_last_error();
        (void)hipGetLastError();
        ehyb_plan_destroy(*plan);
        *plan = nullptr;
        rc = create_host_plan(m, row_begin, row_end, cfg, n_col_segs, col_seg_first, false, plan);
        if (rc != EHYB_OK) return rc;
        if ((*plan)->cfg.verbose) printf("plan: the device builder failed (%s): panel form rebuilt on the host\n", why.c_str());
        on_device = false;
    }
    const double t2 = wall_seconds();
    if (rc == EHYB_OK) rc = ehyb_plan_upload(*plan);
    if (rc != EHYB_OK) {
        ehyb_plan_destroy(*plan);
        *plan = nullptr;
        return rc;
    }
    if ((*plan)->cfg.verbose)
        printf("plan: host layout %.3f s, panel form %s %.3f s, upload %.3f s\n", t1 - t0, on_device ? "on the device" : "(host, included)", t2 - t1,
               wall_seconds() - t2);
    return rc;
}

// The drop-in entry point (reference spmv.cu:61-133).  cfg == NULL: defaults, storage chosen from
// the matrix (sym_storage_suits) -- no environment variable takes part.
int spmvGPuEHYB_cfg(matrixCOO* localMatrix, const double* vectorIn, double* vectorOut, const int MAXIter,
                    int* realIter, const ehyb_config* cfg_in, double* ms_total)
{
    clear_error();
    if (!localMatrix || !vectorIn || !vectorOut || MAXIter < 0)
        EHYB_FAIL(EHYB_ERR_ARG, "spmvGPuEHYB: bad arguments");
    ehyb_config cfg;
    memset(&cfg, 0, sizeof cfg);  // zero = default; the sizes follow from the mode fields
    if (cfg_in)
        cfg = *cfg_in;
    else if (sym_storage_suits(localMatrix))
        cfg.sym_pairs = 1;
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
    // part of the warm-up: the heaviest work items onto the XCDs of THIS device that stream fastest (a few stamped
    // launches; keeps what it finds only if the launch got shorter; its failure is not the multiply's)
    if (ehyb_plan_tune(P, dx, dy, 3, nullptr, nullptr) != EHYB_OK) clear_error();
    rc = ehyb_spmv_bench(P, dx, dy, nullptr, 10, iters, &ms, nullptr, nullptr);  // spmv.cu:100-116
    if (rc != EHYB_OK) return fail(rc);
    if (hipMemcpy(vectorOut, dy, n * 8, hipMemcpyDeviceToHost) != hipSuccess) {
        set_error("spmvGPuEHYB: download of y failed");
        return fail(EHYB_ERR_HIP);
    }
    printf("iter is %d, time is %f ms, GPU Gflops is %f\n ", iters, ms,
           (1e-9 * ((double)localMatrix->totalNum * 2) * 1000 * iters) / ms);  // spmv.cu:121-122
    if (realIter) *realIter = iters;
    if (ms_total) *ms_total = ms;
    fail(EHYB_OK);
    return EHYB_OK;
}

int spmvGPuEHYB_status(matrixCOO* localMatrix, const double* vectorIn, double* vectorOut, const int MAXIter,
                       int* realIter)
{
    return spmvGPuEHYB_cfg(localMatrix, vectorIn, vectorOut, MAXIter, realIter, nullptr, nullptr);
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
