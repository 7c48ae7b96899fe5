// Symbolic phase of the panel residual on the device (the reference does all of COO2EHYB on one host
// thread, convert.c:170-369).
//
// The panel form (er_panel.cpp) is a re-ordering of the residual entries -- by column panel, inside a panel by
// (row, column) -- plus bookkeeping that follows from that order: pieces (runs of one row inside a 64-entry chunk),
// their slots (pieces in row-block order), the compressed slot stream.  On the host that is a counting sort over
// hundreds of millions of entries and several passes over multi-gigabyte arrays whose pages a fresh process has
// never touched: R-MAT 2^24 spent 5 s in it on the 16 CPUs of the GPU box.  Here the same arrays are made where
// they are used:
//
//   row of every entry      row heads scattered through the residual row pointer, inclusive max-scan
//   pass-1 order            ONE stable radix sort of (panel | row | local column) keys with the entry index as
//                           payload (rocPRIM; 48 key bits for R-MAT 2^24)
//   panel starts            binary search of every panel's first key; padded starts, work items and units on the
//                           host (a thousand panels: the host code of er_panel.cpp, shared)
//   streams                 one gather kernel: value, local column, row, source entry to their padded places
//   pieces                  head flags + inclusive sum-scan; row of every piece
//   slots                   stable radix sort of the pieces by row block; inverse permutation
//   compressed slots        one wave per 64-entry chunk: head / jump flags by ballot, jump list through a scan of the
//                           per-chunk counts (encode_panel_slots' definition)
//
// Every array equals the host builder's, entry for entry, where the rows arrive in column order (the reorder step's
// output; a row that does not is sorted here as a whole, on the host per CSR segment: another valid order of the
// same sums) -- tests/test_gpu_symbolic.py compares them.  The sort and the scans are library primitives; the hot path
// (the multiply) has none.
#include <hip/hip_runtime.h>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/functional.hpp>

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

constexpr int kT = 256;
constexpr size_t kSlack = 4096;  // like upload(): the kernels' clamped prefetches may read past the end

struct Dev {  // device memory that goes away with the scope unless taken
    void* p = nullptr;
    Dev() = default;
    Dev(const Dev&) = delete;
    Dev& operator=(const Dev&) = delete;
    ~Dev() { drop(); }
    void drop()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
    }
    hipError_t get(size_t bytes)
    {
        drop();
        return hipMalloc(&p, std::max<size_t>(bytes, 16) + kSlack);
    }
    template <class T>
    T* as() const
    {
        return (T*)p;
    }
    template <class T>
    T* take()
    {
        T* q = (T*)p;
        p = nullptr;
        return q;
    }
};

inline dim3 grid_of(int64_t n) { return dim3((unsigned)std::max<int64_t>(1, (n + kT - 1) / kT)); }

struct Geo {  // what a kernel needs to find the panel of a column
    const int32_t* seg_first;
    const int32_t* seg_panel0;
    int n_segs;
    int W;
};

__device__ inline int panel_of(const Geo& g, int c, int* first_col)
{
    int s = 0;
    while (s + 1 < g.n_segs && c >= g.seg_first[s + 1]) ++s;  // a handful of segments
    const int q = (c - g.seg_first[s]) / g.W;
    *first_col = g.seg_first[s] + q * g.W;
    return g.seg_panel0[s] + q;
}

// rowtag[first entry of row r] = r for every non-empty row (the rest is zero): the max-scan then gives every entry its row
__global__ __launch_bounds__(kT) void row_heads_kernel(const int64_t* __restrict__ rp, int nrows, int32_t* __restrict__ rowtag)
{
    const int r = blockIdx.x * kT + threadIdx.x;
    if (r < nrows && rp[r + 1] > rp[r]) rowtag[rp[r]] = r;
}

__global__ __launch_bounds__(kT) void keys_kernel(const int32_t* __restrict__ col, const int32_t* __restrict__ row, int64_t n, Geo g, int n_cols,
                                                  int row_shift, int panel_shift, uint64_t* __restrict__ key, uint32_t* __restrict__ idx,
                                                  int* __restrict__ bad)
{
    const int64_t k = (int64_t)blockIdx.x * kT + threadIdx.x;
    if (k >= n) return;
    int c = col[k];
    if ((unsigned)c >= (unsigned)n_cols) {
        *bad = 1;
        c = 0;
    }
    int first;
    const int p = panel_of(g, c, &first);
    key[k] = (uint64_t)p << panel_shift | (uint64_t)(uint32_t)row[k] << row_shift | (uint64_t)(c - first);
    idx[k] = (uint32_t)k;
}

// first[i] = number of keys below (i << shift), i = 0 .. count (first[count] = n)
template <class K>
__global__ __launch_bounds__(kT) void bounds_kernel(const K* __restrict__ key, int64_t n, int shift, int count, int64_t* __restrict__ first)
{
    const int i = blockIdx.x * kT + threadIdx.x;
    if (i > count) return;
    if (i == count) {
        first[i] = n;
        return;
    }
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if ((int64_t)(key[mid] >> shift) < i)
            lo = mid + 1;
        else
            hi = mid;
    }
    first[i] = lo;
}

__global__ __launch_bounds__(kT) void deal_kernel(const uint64_t* __restrict__ key, const uint32_t* __restrict__ idx, int64_t n, int row_shift,
                                                  int panel_shift, const int64_t* __restrict__ pcount, const int64_t* __restrict__ pstart,
                                                  const double* __restrict__ val, const int32_t* __restrict__ src, int64_t src_base,
                                                  double* __restrict__ pb_val, uint16_t* __restrict__ pb_col, int32_t* __restrict__ prow,
                                                  int32_t* __restrict__ pb_src)
{
    const int64_t i = (int64_t)blockIdx.x * kT + threadIdx.x;
    if (i >= n) return;
    const uint64_t q = key[i];
    const int p = (int)(q >> panel_shift);
    const int64_t pos = pstart[p] + (i - pcount[p]);
    const uint32_t k = idx[i];
    pb_val[pos] = val[k];
    pb_col[pos] = (uint16_t)(q & ((1u << row_shift) - 1));
    prow[pos] = (int32_t)((q >> row_shift) & ((1ull << (panel_shift - row_shift)) - 1));
    if (pb_src) pb_src[pos] = src ? src[k] : (int32_t)(src_base + k);
}

// a piece begins at every stored entry that opens a 64-entry chunk or follows an entry of another row
__global__ __launch_bounds__(kT) void heads_kernel(const int32_t* __restrict__ prow, int64_t padded, uint32_t* __restrict__ head)
{
    const int64_t pos = (int64_t)blockIdx.x * kT + threadIdx.x;
    if (pos >= padded) return;
    const int32_t r = prow[pos];
    head[pos] = (r >= 0 && ((pos & 63) == 0 || prow[pos - 1] != r)) ? 1u : 0u;
}

__global__ __launch_bounds__(kT) void piece_rows_kernel(const int32_t* __restrict__ prow, const uint32_t* __restrict__ head,
                                                        const uint32_t* __restrict__ pinc, int64_t padded, const int32_t* __restrict__ rb_first,
                                                        int n_rb, int32_t* __restrict__ piece_row, uint32_t* __restrict__ piece_rb,
                                                        uint32_t* __restrict__ piece_idx)
{
    const int64_t pos = (int64_t)blockIdx.x * kT + threadIdx.x;
    if (pos >= padded || !head[pos]) return;
    const uint32_t j = pinc[pos] - 1;
    const int32_t r = prow[pos];
    int lo = 0, hi = n_rb;  // last block whose first row is <= r
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (rb_first[mid] <= r)
            lo = mid;
        else
            hi = mid;
    }
    piece_row[j] = r;
    piece_rb[j] = (uint32_t)lo;
    piece_idx[j] = j;
}

__global__ __launch_bounds__(kT) void slots_kernel(const uint32_t* __restrict__ rb_sorted, const uint32_t* __restrict__ piece_sorted, int64_t n_pieces,
                                                   const int32_t* __restrict__ piece_row, const int32_t* __restrict__ rb_first,
                                                   uint32_t* __restrict__ slot_of_piece, uint16_t* __restrict__ pb_row)
{
    const int64_t s = (int64_t)blockIdx.x * kT + threadIdx.x;
    if (s >= n_pieces) return;
    const uint32_t j = piece_sorted[s];
    slot_of_piece[j] = (uint32_t)s;
    pb_row[s] = (uint16_t)(piece_row[j] - rb_first[rb_sorted[s]]);
}

__global__ __launch_bounds__(kT) void dst_kernel(const int32_t* __restrict__ prow, const uint32_t* __restrict__ pinc,
                                                 const uint32_t* __restrict__ slot_of_piece, int64_t padded, uint32_t* __restrict__ dst)
{
    const int64_t pos = (int64_t)blockIdx.x * kT + threadIdx.x;
    if (pos >= padded) return;
    dst[pos] = prow[pos] >= 0 ? slot_of_piece[pinc[pos] - 1] : 0xFFFFFFFFu;
}

// encode_panel_slots on the device: one wave per chunk.  FILL = false counts the chunk's jumps, FILL = true writes the
// column words with their flags and the jump list (chunk_first = exclusive scan of the counts).
template <bool FILL>
__global__ __launch_bounds__(kT) void encode_kernel(const uint32_t* __restrict__ dst, const uint16_t* __restrict__ col, int64_t chunks,
                                                    uint32_t* __restrict__ chunk_jumps, const uint32_t* __restrict__ chunk_first,
                                                    uint16_t* __restrict__ colf, uint32_t* __restrict__ jump)
{
    const int lane = threadIdx.x & 63;
    const int64_t c = (int64_t)blockIdx.x * (kT / 64) + (threadIdx.x >> 6);
    if (c >= chunks) return;  // (whole waves leave together)
    const uint32_t d = dst[c * 64 + lane];
    const uint32_t dprev = __shfl_up(d, 1);
    const bool head = lane == 0 || d != dprev;
    const bool jmp = lane == 0 || (head && d != dprev + 1);
    const unsigned long long headmask = __ballot(head), jumpmask = __ballot(jmp);
    if (!FILL) {
        if (lane == 0) chunk_jumps[c] = (uint32_t)__popcll(jumpmask);
        return;
    }
    const unsigned long long below = (1ull << lane) - 1;
    colf[c * 64 + lane] = (uint16_t)(col[c * 64 + lane] | (head ? 0x8000u : 0u) | (jmp ? 0x4000u : 0u));
    if (jmp) {
        const uint32_t pieces = (uint32_t)__popcll(headmask & (below | (1ull << lane))) - 1;  // pieces begun before this one in the chunk
        jump[chunk_first[c] + (uint32_t)__popcll(jumpmask & below)] = d - pieces;
    }
}

int bits_for(int64_t count)  // bits that hold 0 .. count - 1
{
    int b = 1;
    while (((int64_t)1 << b) < count) ++b;
    return b;
}

struct Laps {
    bool on;
    double t;
    explicit Laps(bool v) : on(v), t(wall_seconds()) {}
    void lap(const char* what)
    {
        if (!on) return;
        (void)hipDeviceSynchronize();
        const double now = wall_seconds();
        printf("  panel form on the device: %-28s %7.1f ms\n", what, (now - t) * 1e3);
        t = now;
    }
};

template <class T>
int to_dev(Dev* d, const T* host, size_t n)
{
    HIP_TRY(d->get(n * sizeof(T)));
    if (n) HIP_TRY(hipMemcpy(d->p, host, n * sizeof(T), hipMemcpyHostToDevice));
    return EHYB_OK;
}

}  // namespace

int ehyb::build_panel_on_device(ehyb_plan* P)
{
    HostLayout& L = P->host;
    HostLayout::Deferred& D = L.deferred;
    if (!D.pending) return EHYB_OK;
    const Config& cfg = P->cfg;
    const int64_t n = D.nnz_er;
    const int nrows = L.row_end - L.row_begin;
    if (n <= 0 || (int64_t)D.er_rp.size() != (int64_t)nrows + 1 || D.er_rp[(size_t)nrows] != n || !D.col || !D.val)
        EHYB_FAIL(EHYB_ERR_INTERNAL, "build_panel_on_device: nothing consistent was left to build");
    Laps laps(cfg.verbose > 1);

    // ---- host: row blocks and panels (the entries per row are the row pointer's differences)
    PanelGeometry G;
    {
        std::vector<int32_t> cnt_row((size_t)nrows);
#pragma omp parallel for schedule(static, 65536)
        for (int r = 0; r < nrows; ++r) cnt_row[(size_t)r] = (int32_t)(D.er_rp[(size_t)r + 1] - D.er_rp[(size_t)r]);
        const int rc = panel_geometry(cfg, L, cnt_row.data(), n, &G);
        if (rc != EHYB_OK) return rc;
    }
    const int W = G.W;
    const int n_segs = (int)G.seg_first.size() - 1;
    const int n_panels = (int)G.panel_first.size() - 1;
    const int n_rb = (int)G.rb_first.size() - 1;
    if (W > 16384) EHYB_FAIL(EHYB_ERR_ARG, "build_panel_on_device: panels of %d columns (14 bits hold the local column)", W);
    const int row_shift = 14, panel_shift = row_shift + bits_for(nrows), key_bits = panel_shift + bits_for(n_panels);
    if (key_bits > 64) EHYB_FAIL(EHYB_ERR_ARG, "build_panel_on_device: sort key of %d bits", key_bits);

    // ---- geometry and the entries to the device
    Dev d_seg_first, d_seg_panel0, d_rb_first;
    int rc;
    if ((rc = to_dev(&d_seg_first, G.seg_first.data(), G.seg_first.size())) != EHYB_OK) return rc;
    if ((rc = to_dev(&d_seg_panel0, G.seg_panel0.data(), G.seg_panel0.size())) != EHYB_OK) return rc;
    {
        std::vector<int32_t> rel(G.rb_first);  // rows are counted from the plan's first row on the device
        for (int32_t& v : rel) v -= L.row_begin;
        if ((rc = to_dev(&d_rb_first, rel.data(), rel.size())) != EHYB_OK) return rc;
    }
    const Geo geo = {d_seg_first.as<int32_t>(), d_seg_panel0.as<int32_t>(), n_segs, W};
    Dev d_key[2], d_idx[2], d_tmp, d_bad;
    HIP_TRY(d_bad.get(sizeof(int)));
    HIP_TRY(hipMemset(d_bad.p, 0, sizeof(int)));
    for (int b = 0; b < 2; ++b) {
        HIP_TRY(d_key[b].get((size_t)n * 8));
        HIP_TRY(d_idx[b].get((size_t)n * 4));
    }
    {
        Dev d_col, d_rp, d_tag, d_row;
        if ((rc = to_dev(&d_col, D.col, (size_t)n)) != EHYB_OK) return rc;
        if ((rc = to_dev(&d_rp, D.er_rp.data(), D.er_rp.size())) != EHYB_OK) return rc;
        laps.lap("columns + row pointer uploaded");
        HIP_TRY(d_tag.get((size_t)n * 4));
        HIP_TRY(d_row.get((size_t)n * 4));
        HIP_TRY(hipMemset(d_tag.p, 0, (size_t)n * 4));
        hipLaunchKernelGGL(row_heads_kernel, grid_of(nrows), dim3(kT), 0, 0, d_rp.as<int64_t>(), nrows, d_tag.as<int32_t>());
        HIP_TRY(hipGetLastError());
        size_t tmp_bytes = 0;
        HIP_TRY(rocprim::inclusive_scan(nullptr, tmp_bytes, d_tag.as<int32_t>(), d_row.as<int32_t>(), (size_t)n, rocprim::maximum<int32_t>()));
        HIP_TRY(d_tmp.get(tmp_bytes));
        HIP_TRY(rocprim::inclusive_scan(d_tmp.p, tmp_bytes, d_tag.as<int32_t>(), d_row.as<int32_t>(), (size_t)n, rocprim::maximum<int32_t>()));
        hipLaunchKernelGGL(keys_kernel, grid_of(n), dim3(kT), 0, 0, d_col.as<int32_t>(), d_row.as<int32_t>(), n, geo, L.n_cols, row_shift, panel_shift,
                           d_key[0].as<uint64_t>(), d_idx[0].as<uint32_t>(), d_bad.as<int>());
        HIP_TRY(hipGetLastError());
        int bad = 0;
        HIP_TRY(hipMemcpy(&bad, d_bad.p, sizeof(int), hipMemcpyDeviceToHost));
        if (bad) EHYB_FAIL(EHYB_ERR_INTERNAL, "build_panel_on_device: column outside the matrix");
    }
    laps.lap("rows and keys");
    // ---- pass-1 order: stable sort by (panel, row, local column)
    rocprim::double_buffer<uint64_t> keys(d_key[0].as<uint64_t>(), d_key[1].as<uint64_t>());
    rocprim::double_buffer<uint32_t> idxs(d_idx[0].as<uint32_t>(), d_idx[1].as<uint32_t>());
    {
        size_t tmp_bytes = 0;
        HIP_TRY(rocprim::radix_sort_pairs(nullptr, tmp_bytes, keys, idxs, (size_t)n, 0u, (unsigned)key_bits));
        HIP_TRY(d_tmp.get(tmp_bytes));
        HIP_TRY(rocprim::radix_sort_pairs(d_tmp.p, tmp_bytes, keys, idxs, (size_t)n, 0u, (unsigned)key_bits));
    }
    laps.lap("radix sort of the entries");
    // ---- panels: where each starts; padded starts (multiples of 64 entries) and the work of pass 1 on the host
    std::vector<int64_t> pcount((size_t)n_panels + 1), pstart((size_t)n_panels + 1, 0);
    Dev d_pcount, d_pstart;
    HIP_TRY(d_pcount.get(pcount.size() * 8));
    hipLaunchKernelGGL(bounds_kernel<uint64_t>, grid_of(n_panels + 1), dim3(kT), 0, 0, keys.current(), n, panel_shift, n_panels, d_pcount.as<int64_t>());
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(pcount.data(), d_pcount.p, pcount.size() * 8, hipMemcpyDeviceToHost));
    for (int p = 0; p < n_panels; ++p) pstart[(size_t)p + 1] = pstart[(size_t)p] + (pcount[(size_t)p + 1] - pcount[(size_t)p] + 63) / 64 * 64;
    const int64_t padded = pstart[(size_t)n_panels];
    const int64_t chunks = padded / 64;
    if (padded >= 0x7FFFFF00ll) EHYB_FAIL(EHYB_ERR_ARG, "build_panel_on_device: %lld padded entries do not fit 32-bit offsets", (long long)padded);
    if ((rc = to_dev(&d_pstart, pstart.data(), pstart.size())) != EHYB_OK) return rc;
    const int64_t staged = panel_pass1_items(cfg, G, pstart, n, &L);
    // ---- the streams
    Dev d_val, d_src, d_pb_val, d_pb_col, d_prow, d_pb_src;
    if ((rc = to_dev(&d_val, D.val, (size_t)n)) != EHYB_OK) return rc;
    if (D.src && (rc = to_dev(&d_src, D.src, (size_t)n)) != EHYB_OK) return rc;
    laps.lap("values uploaded");
    HIP_TRY(d_pb_val.get((size_t)padded * 8));
    HIP_TRY(d_pb_col.get((size_t)padded * 2));
    HIP_TRY(d_prow.get((size_t)padded * 4));
    HIP_TRY(hipMemset(d_pb_val.p, 0, (size_t)padded * 8 + kSlack));
    HIP_TRY(hipMemset(d_pb_col.p, 0, (size_t)padded * 2 + kSlack));
    HIP_TRY(hipMemset(d_prow.p, 0xFF, (size_t)padded * 4));
    if (D.want_src) {
        HIP_TRY(d_pb_src.get((size_t)padded * 4));
        HIP_TRY(hipMemset(d_pb_src.p, 0xFF, (size_t)padded * 4));
    }
    hipLaunchKernelGGL(deal_kernel, grid_of(n), dim3(kT), 0, 0, keys.current(), idxs.current(), n, row_shift, panel_shift, d_pcount.as<int64_t>(),
                       d_pstart.as<int64_t>(), d_val.as<double>(), D.src ? d_src.as<int32_t>() : nullptr, D.src_base, d_pb_val.as<double>(),
                       d_pb_col.as<uint16_t>(), d_prow.as<int32_t>(), D.want_src ? d_pb_src.as<int32_t>() : nullptr);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    d_val.drop();
    d_src.drop();
    for (int b = 0; b < 2; ++b) {
        d_key[b].drop();
        d_idx[b].drop();
    }
    laps.lap("entries dealt to the panels");
    // ---- pieces
    Dev d_head, d_pinc;
    HIP_TRY(d_head.get((size_t)padded * 4));
    HIP_TRY(d_pinc.get((size_t)padded * 4));
    hipLaunchKernelGGL(heads_kernel, grid_of(padded), dim3(kT), 0, 0, d_prow.as<int32_t>(), padded, d_head.as<uint32_t>());
    HIP_TRY(hipGetLastError());
    {
        size_t tmp_bytes = 0;
        HIP_TRY(rocprim::inclusive_scan(nullptr, tmp_bytes, d_head.as<uint32_t>(), d_pinc.as<uint32_t>(), (size_t)padded, rocprim::plus<uint32_t>()));
        HIP_TRY(d_tmp.get(tmp_bytes));
        HIP_TRY(rocprim::inclusive_scan(d_tmp.p, tmp_bytes, d_head.as<uint32_t>(), d_pinc.as<uint32_t>(), (size_t)padded, rocprim::plus<uint32_t>()));
    }
    uint32_t last = 0;
    HIP_TRY(hipMemcpy(&last, d_pinc.as<uint32_t>() + (padded - 1), 4, hipMemcpyDeviceToHost));
    const int64_t n_pieces = (int64_t)last;
    if (n_pieces <= 0 || n_pieces >= 0x7FFFFF00ll) EHYB_FAIL(EHYB_ERR_INTERNAL, "build_panel_on_device: %lld partial sums", (long long)n_pieces);
    // ---- slots: the pieces in (row block, pass-1 order)
    Dev d_piece_row, d_rb[2], d_pc[2], d_slot, d_pb_row, d_rbcount;
    HIP_TRY(d_piece_row.get((size_t)n_pieces * 4));
    for (int b = 0; b < 2; ++b) {
        HIP_TRY(d_rb[b].get((size_t)n_pieces * 4));
        HIP_TRY(d_pc[b].get((size_t)n_pieces * 4));
    }
    hipLaunchKernelGGL(piece_rows_kernel, grid_of(padded), dim3(kT), 0, 0, d_prow.as<int32_t>(), d_head.as<uint32_t>(), d_pinc.as<uint32_t>(), padded,
                       d_rb_first.as<int32_t>(), n_rb, d_piece_row.as<int32_t>(), d_rb[0].as<uint32_t>(), d_pc[0].as<uint32_t>());
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    d_head.drop();
    rocprim::double_buffer<uint32_t> rbk(d_rb[0].as<uint32_t>(), d_rb[1].as<uint32_t>());
    rocprim::double_buffer<uint32_t> pcs(d_pc[0].as<uint32_t>(), d_pc[1].as<uint32_t>());
    {
        size_t tmp_bytes = 0;
        const unsigned rb_bits = (unsigned)bits_for(n_rb);
        HIP_TRY(rocprim::radix_sort_pairs(nullptr, tmp_bytes, rbk, pcs, (size_t)n_pieces, 0u, rb_bits));
        HIP_TRY(d_tmp.get(tmp_bytes));
        HIP_TRY(rocprim::radix_sort_pairs(d_tmp.p, tmp_bytes, rbk, pcs, (size_t)n_pieces, 0u, rb_bits));
    }
    std::vector<int64_t> rb_count((size_t)n_rb + 1);
    HIP_TRY(d_rbcount.get(rb_count.size() * 8));
    hipLaunchKernelGGL(bounds_kernel<uint32_t>, grid_of(n_rb + 1), dim3(kT), 0, 0, rbk.current(), n_pieces, 0, n_rb, d_rbcount.as<int64_t>());
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(rb_count.data(), d_rbcount.p, rb_count.size() * 8, hipMemcpyDeviceToHost));
    HIP_TRY(d_slot.get((size_t)n_pieces * 4));
    HIP_TRY(d_pb_row.get((size_t)n_pieces * 2));
    hipLaunchKernelGGL(slots_kernel, grid_of(n_pieces), dim3(kT), 0, 0, rbk.current(), pcs.current(), n_pieces, d_piece_row.as<int32_t>(),
                       d_rb_first.as<int32_t>(), d_slot.as<uint32_t>(), d_pb_row.as<uint16_t>());
    HIP_TRY(hipGetLastError());
    Dev d_dst;
    HIP_TRY(d_dst.get((size_t)padded * 4));
    hipLaunchKernelGGL(dst_kernel, grid_of(padded), dim3(kT), 0, 0, d_prow.as<int32_t>(), d_pinc.as<uint32_t>(), d_slot.as<uint32_t>(), padded,
                       d_dst.as<uint32_t>());
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    d_prow.drop();
    d_pinc.drop();
    d_slot.drop();
    d_piece_row.drop();
    for (int b = 0; b < 2; ++b) {
        d_rb[b].drop();
        d_pc[b].drop();
    }
    laps.lap("pieces and slots");
    // ---- what pass 1 streams instead of a slot per entry
    Dev d_cj, d_chunk, d_colf, d_jump;
    HIP_TRY(d_cj.get((size_t)(chunks + 1) * 4));
    HIP_TRY(d_chunk.get((size_t)(chunks + 1) * 4));
    HIP_TRY(hipMemset(d_cj.p, 0, (size_t)(chunks + 1) * 4));
    const dim3 egrid((unsigned)((chunks + kT / 64 - 1) / (kT / 64)));
    hipLaunchKernelGGL(encode_kernel<false>, egrid, dim3(kT), 0, 0, d_dst.as<uint32_t>(), d_pb_col.as<uint16_t>(), chunks, d_cj.as<uint32_t>(), nullptr, nullptr,
                       nullptr);
    HIP_TRY(hipGetLastError());
    {
        size_t tmp_bytes = 0;
        HIP_TRY(rocprim::exclusive_scan(nullptr, tmp_bytes, d_cj.as<uint32_t>(), d_chunk.as<uint32_t>(), 0u, (size_t)chunks + 1, rocprim::plus<uint32_t>()));
        HIP_TRY(d_tmp.get(tmp_bytes));
        HIP_TRY(rocprim::exclusive_scan(d_tmp.p, tmp_bytes, d_cj.as<uint32_t>(), d_chunk.as<uint32_t>(), 0u, (size_t)chunks + 1, rocprim::plus<uint32_t>()));
    }
    uint32_t n_jumps = 0;
    HIP_TRY(hipMemcpy(&n_jumps, d_chunk.as<uint32_t>() + chunks, 4, hipMemcpyDeviceToHost));
    HIP_TRY(d_colf.get((size_t)padded * 2));
    HIP_TRY(d_jump.get((size_t)n_jumps * 4));
    HIP_TRY(hipMemset(d_colf.as<char>() + (size_t)padded * 2, 0, kSlack));
    hipLaunchKernelGGL(encode_kernel<true>, egrid, dim3(kT), 0, 0, d_dst.as<uint32_t>(), d_pb_col.as<uint16_t>(), chunks, nullptr, d_chunk.as<uint32_t>(),
                       d_colf.as<uint16_t>(), d_jump.as<uint32_t>());
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    laps.lap("compressed slots");
    // ---- host: pass-2 units, the scalars of the form; the small arrays go up with ehyb_plan_upload
    panel_finish(G, rb_count, staged, padded, n_pieces, (int64_t)n_jumps, &L);
    L.stats.er_partials = L.pb_partials;
    L.stats.bytes_format = L.stats.bytes_format_ell + L.pb_bytes;
    P->d_pb_val = d_pb_val.take<double>();
    P->d_pb_colf = d_colf.take<uint16_t>();
    P->d_pb_chunk = d_chunk.take<uint32_t>();
    P->d_pb_jump = d_jump.take<uint32_t>();
    P->d_pb_row = d_pb_row.take<uint16_t>();
    if (D.want_src) P->d_pb_src = d_pb_src.take<int32_t>();
    L.pb_host_missing = true;
    L.deferred = HostLayout::Deferred();  // the row-order copies (if any) go back
    if (cfg.verbose)
        printf("panel residual (built on the device): %lld entries (%lld with padding) in %d panels of %d columns -> %lld partials, %zu + %zu work units, "
               "row blocks <= %d rows\n",
               (long long)n, (long long)padded, n_panels, W, (long long)n_pieces, L.pb_items1.size() / 2, L.pb_units2.size() / 4, L.pb_rows_max);
    return EHYB_OK;
}

// The streams of a device-built panel form, back on the host for whoever reads the plan's arrays (ehyb_plan_host_array,
// ehyb_plan_save): value, column + flags, chunk records, jump list, rows of the partials, slot map; the plain column and
// slot arrays (pb_col, pb_dst: the definition the layout tests read) are decoded from them.
int ehyb::materialize_panel_host(ehyb_plan* P)
{
    HostLayout& L = P->host;
    if (!L.pb_host_missing) return EHYB_OK;
    if (!P->d_pb_val || !P->d_pb_colf || !P->d_pb_chunk || !P->d_pb_jump || !P->d_pb_row)
        EHYB_FAIL(EHYB_ERR_STATE, "materialize_panel_host: the device arrays of the panel form are gone");
    const int64_t padded = L.pb_padded, chunks = padded / 64;
    try {
        L.pb_val.resize((size_t)padded);
        L.pb_colf.resize((size_t)padded);
        L.pb_chunk.resize((size_t)chunks + 1);
        L.pb_row.resize((size_t)L.pb_partials);
        HIP_TRY(hipMemcpy(L.pb_val.data(), P->d_pb_val, (size_t)padded * 8, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(L.pb_colf.data(), P->d_pb_colf, (size_t)padded * 2, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(L.pb_chunk.data(), P->d_pb_chunk, ((size_t)chunks + 1) * 4, hipMemcpyDeviceToHost));
        L.pb_jump.resize((size_t)L.pb_chunk[(size_t)chunks]);
        if (!L.pb_jump.empty()) HIP_TRY(hipMemcpy(L.pb_jump.data(), P->d_pb_jump, L.pb_jump.size() * 4, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(L.pb_row.data(), P->d_pb_row, (size_t)L.pb_partials * 2, hipMemcpyDeviceToHost));
        if (P->d_pb_src) {
            L.pb_src.resize((size_t)padded);
            HIP_TRY(hipMemcpy(L.pb_src.data(), P->d_pb_src, (size_t)padded * 4, hipMemcpyDeviceToHost));
        }
        L.pb_col.resize((size_t)padded);
        L.pb_dst.resize((size_t)padded);
#pragma omp parallel for schedule(static, 1024)
        for (int64_t c = 0; c < chunks; ++c) {
            uint32_t j = L.pb_chunk[(size_t)c], base = 0, pieces = 0;
            for (int l = 0; l < 64; ++l) {
                const uint16_t w = L.pb_colf[(size_t)(c * 64 + l)];
                if (l > 0 && (w & 0x8000u)) ++pieces;
                if (w & 0x4000u) base = L.pb_jump[(size_t)j++];
                L.pb_col[(size_t)(c * 64 + l)] = (uint16_t)(w & 0x3FFFu);
                L.pb_dst[(size_t)(c * 64 + l)] = base + pieces;
            }
        }
    } catch (const std::bad_alloc&) {
        EHYB_FAIL(EHYB_ERR_ALLOC, "materialize_panel_host: out of host memory");
    }
    L.pb_host_missing = false;
    return EHYB_OK;
}
