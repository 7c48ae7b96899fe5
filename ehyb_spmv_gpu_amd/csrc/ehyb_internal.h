// Internal declarations shared by the host builder, the partitioner and the HIP side.
// Nothing here is part of the C-ABI (see include/ehyb.h).
#pragma once
#include <atomic>
#include <cstdint>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <memory>
#include <new>
#include <string>
#include <utility>
#include <vector>
#include "ehyb.h"

namespace ehyb {

// The big arrays of a layout (hundreds of MB each): a vector whose resize() does NOT write the new elements -- the builder
// fills every element itself, partition by partition on all host threads; a value-initialising resize was one more pass
// over the array on ONE thread (audikw_1-like: 1.2 GB of such fills, a quarter of the plan's build time).
// release_big: free() of a malloc()ed block; a block of 64 MiB or more is handed to a helper thread -- giving 600 MB of touched pages back to the
// kernel takes 70-75 ms on the GPU box's host (more than gathering them took), and nothing the caller does next has to wait for it.
void release_big(void* p, size_t bytes);
template <class T>
struct NoInitAlloc {
    using value_type = T;
    template <class U>
    struct rebind {
        using other = NoInitAlloc<U>;
    };
    NoInitAlloc() = default;
    template <class U>
    NoInitAlloc(const NoInitAlloc<U>&) {}
    T* allocate(size_t n)
    {
        void* p = malloc(std::max<size_t>(n * sizeof(T), 1));
        if (!p) throw std::bad_alloc();
        return static_cast<T*>(p);
    }
    void deallocate(T* p, size_t) { free(p); }
    template <class U>
    bool operator==(const NoInitAlloc<U>&) const { return true; }
    template <class U>
    bool operator!=(const NoInitAlloc<U>&) const { return false; }
    template <class U, class... A>
    void construct(U* p, A&&... a)
    {
        if constexpr (sizeof...(A) == 0)
            ::new ((void*)p) U;  // default-initialised: left as it is for arithmetic types
        else
            ::new ((void*)p) U(std::forward<A>(a)...);
    }
};
template <class T>
using BigVec = std::vector<T, NoInitAlloc<T>>;

// ---------------------------------------------------------------- errors
void set_error(const char* fmt, ...);
void clear_error();

#define EHYB_FAIL(code, ...)            \
    do {                                \
        ::ehyb::set_error(__VA_ARGS__); \
        return (code);                  \
    } while (0)

// ---------------------------------------------------------------- config
// Resolved configuration: every field has its final value.
struct Config {
    int lds_doubles;
    int part_rows;
    int threads;
    int window_mode;
    int items_per_cu;
    int partitioner;
    int er_seg_len;
    int host_threads;
    int verbose;
    int seed;
    int n_top;
    int er_threads;
    int ell_variant;
    int col_sharing;
    int fuse_er;
    int cap_split;
    int hub_rule;
    int sym_pairs;
    int part_boundary_cap;  // ints the caller's partBoundary holds; 0 = m->nParts + 1, never more
    int er_mode;            // 0 automatic, 1 CSR segments, 2 panel form
    int er_panel_cols;
    int er_block_rows;
    int direct;             // 0 automatic, 1 on, 2 off
    int ell_prune;          // 0/1: windows that cost more than the panel residual are given up, 2 = never
    int value_map;          // 1: keep the slot maps of the value streams (ehyb_plan_set_values)
    int prune_pct;          // ell_prune threshold in per cent of the panel form's cost (110)
    int er_units1;          // panel form: work units aimed at, pass 1 / pass 2 (2048)
    int er_units2;
    int graph_compress;     // 0 automatic (on with symmetric pair storage), 1 on, 2 off, 3 on + a refinement on the rows themselves (twins may part)
    int balance;            // 0 automatic, 1 entries, 2 rows
    int req_margin;         // 0 default, -1 none, > 0 as given
    int sym_slack_permille; // 30
    int xcd_map;            // 1 on, 2 off
    int graphs;             // 1 on, 2 off
    int er_sums;            // 1 DPP scan, 2 LDS words
    int er_panel_threads;   // 0 automatic, 512, 1024
    int er_queue;           // 0 automatic (queues from six items per resident workgroup up), 1 per-XCD work queues with stealing, 2 one workgroup per item
    int symbolic;           // where the panel form is built by ehyb_plan_create[_segs]: 1 host, 2 device (default)
    int cg_fused_dot;       // 1 on (default), 2 off
    int ell_alternate;      // 0 automatic (streams that do not fit the Infinity Cache), 1 on, 2 off
    int row_split;          // panel form: no row block of pass 2 straddles this row (0 = none)
    int col_map;            // host builder: 1 per-thread column look-up arrays where they fit (default), 2 sorted lists + binary search
    int er_nt;              // panel form, pass 2 reads past the caches: 0 by size, 1 always, 2 never
    int ell_nt;             // window kernel, value stream past the caches: 3 all but the end of an alternating walk (default), 1 every slab, 2 never
};
Config resolve_config(const ehyb_config* cfg);

constexpr int kSlabRows = EHYB_SLAB_ROWS;  // one row per lane
constexpr int kNumCU = 256;                // MI355X: 8 XCD x 32 CU
constexpr int kErBins = 4;                 // lanes per residual segment: 4, 16, 64, 64(atomic)

// ---------------------------------------------------------------- layout
// Host image of the EHYB layout.  Device arrays mirror these one to one.
struct HostLayout {
    int n_cols = 0;       // dimension of the matrix (length of x)
    int row_begin = 0;    // rows covered: [row_begin, row_end)
    int row_end = 0;
    int n_parts = 0;
    int lds_doubles = 0;  // max over partitions of win_len + halo_len

    // per partition
    std::vector<int32_t> part_boundary;  // [n_parts+1] first row
    std::vector<int32_t> win_len;        // [n_parts] contiguous window length (from part start)
    std::vector<int32_t> halo_ptr;       // [n_parts+1]
    std::vector<int32_t> halo_cols;      // gathered columns (global ids), ascending per part
    std::vector<int64_t> part_nnz_ell;   // [n_parts] stored ELL entries of the partition without padding (host only)

    // per slab (64 rows, one wave)
    std::vector<uint32_t> slab_pair_ptr;  // [n_slabs+1] prefix of (width/2)
    std::vector<int32_t> slab_row;        // [n_slabs]
    std::vector<int32_t> slab_part;       // [n_slabs]

    // ELL payload: element (pair p, lane l, half h) at ((pair_ptr[s]+p)*64 + l)*2 + h
    // Column indices are stored once per *group* of lanes with identical column lists:
    // word (pair p, group g) at slab_col_ptr[s] + p*G_s + g holds two 16-bit window-local columns.
    BigVec<double> ell_val;
    BigVec<uint32_t> ell_col;
    std::vector<uint32_t> slab_col_ptr;  // [n_slabs+1] prefix of pairs*groups
    std::vector<uint8_t> lane_group;     // [n_slabs*64]
    std::vector<uint32_t> slab_meta;     // [n_slabs*4] {pair_ptr, col_ptr, first row, pairs<<16 | er_pairs<<8 | groups-1}

    // work items {seg_begin, seg_end, slab_begin, slab_end, er_begin, er_b64, er_b16, er_end}: a run of
    // slabs of (nearly) equal cost; it is cut into segments where it crosses a partition boundary
    std::vector<int32_t> items;
    // segments {partition, slab_begin, slab_end, halo_count, first row, end row, win_len, halo_begin}
    std::vector<int32_t> segs;

    // residual (CSR segments sorted by length, descending)
    std::vector<int64_t> er_seg_ptr;  // [n_seg+1]
    std::vector<int32_t> er_seg_row;  // [n_seg] row | 0x80000000 if the row is split
    std::vector<int32_t> er_col;
    std::vector<double> er_val;
    int32_t er_bins[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // [3] = number of segments
    std::vector<int32_t> er_blocks;                 // {seg_lo, seg_hi, lanes per segment, 0} per block
    // inline form: the residual is also stored as extra pairs behind each slab's ELL pairs
    // (slab_meta word 3, bits 8..15) and the ELL launch multiplies it
    bool inline_er = false;
    // symmetric pair storage: bit 15 of a 16-bit column = "also add value * x[row] to row `column`";
    // the accumulators of a partition's rows live in LDS behind the window (yacc_doubles of them)
    bool sym = false;
    int yacc_doubles = 0;
    std::vector<uint16_t> slab_lrow;  // [n_slabs*64] sym only: image-local row of every lane (0xFFFF: none)

    // direct shape (small matrices): every row in the residual, the residual kernel ASSIGNS y, no ELL launch
    bool direct = false;

    // panel form of a large residual (er_panel.cpp): two streaming passes instead of gathers from global memory
    bool er_panel = false;
    int pb_panel_cols = 0;            // columns per x panel (LDS of pass 1)
    int pb_rows_max = 0;              // rows of the largest row block (LDS of pass 2)
    int64_t pb_partials = 0;
    int64_t pb_bytes = 0;             // bytes both passes move per multiply
    int64_t pb_padded = 0;            // length of pb_val / pb_col / pb_dst / pb_src (panels padded to multiples of 64 entries)
    std::vector<double> pb_val;       // pass-1 order, panels padded to multiples of 64 entries
    std::vector<uint16_t> pb_col;     // column - panel start
    std::vector<uint32_t> pb_dst;     // partial slot; 0xFFFFFFFF = padding
    std::vector<int32_t> pb_units1;   // {first column, columns, first entry, end entry}: a stretch of one panel's entries
    std::vector<int32_t> pb_items1;   // {first unit, end unit}: one pass-1 workgroup's work, a run of units of equal total cost
    std::vector<uint16_t> pb_row;     // per partial: row - first row of its block
    std::vector<int32_t> pb_units2;   // {first partial, end partial, first row, rows}; rows < 0: the block ASSIGNS y (-rows rows)
    // Partitions that went to the residual whole (their window did not pay, plan.cpp) have no ELL work at all:
    // with pb_assign their rows get y from pass 2 alone (row blocks that assign instead of add, also where no
    // partial arrives), the ELL launch skips them, and their slabs cost its work items nothing.
    bool pb_assign = false;
    std::vector<uint8_t> part_windowless;  // [n_parts] host only
    // Column segments (multi-GPU, ehyb_plan_create_host_segs): [0] = 0 < ... < [n] = n_cols; a panel never straddles a
    // boundary, and the pass-1 items of segment s are pb_seg_item[s] .. pb_seg_item[s+1] -- the multiply can then run
    // segment by segment as the x entries of each arrive (ehyb_spmv_part).  Empty = one segment.
    std::vector<int32_t> col_seg_first;
    std::vector<int32_t> pb_seg_item;
    // what pass 1 streams in place of pb_col + pb_dst (derived from them by encode_panel_slots, not stored in plan files)
    std::vector<uint16_t> pb_colf;    // column | head flag (bit 15) | jump flag (bit 14)
    std::vector<uint32_t> pb_chunk;   // per 64-entry chunk: its first jump
    std::vector<uint32_t> pb_jump;    // per jump: slot - pieces before it in its chunk

    // slot maps (cfg.value_map): entry of the source matrix every slot of a value stream was filled from, -1 = padding
    BigVec<int32_t> ell_src;          // like ell_val
    BigVec<int32_t> ell_src2;            // like ell_val, sym only: the mirror entry the slot also stands for, else -1
    std::vector<int32_t> er_src;      // like er_val
    std::vector<int32_t> pb_src;      // like pb_val
    int64_t src_entries = 0;          // entries of the source matrix (length ehyb_plan_set_values expects)

    // Panel form left to the device (ehyb_plan_create with cfg.symbolic = 2, er_panel_dev.hip): the residual entries in
    // ROW order, as the layout builder met them -- either views into the caller's matrix (every entry is residual: R-MAT)
    // or the builder's own row-order copies.  `pending` until the device has dealt them out; after that the pb_*
    // streams live on the device alone until somebody asks for them (pb_host_missing, materialize_panel_host).
    struct Deferred {
        bool pending = false;
        int64_t nnz_er = 0;
        std::vector<int64_t> er_rp;   // [rows + 1] first residual entry of every row
        const int32_t* col = nullptr;
        const double* val = nullptr;
        const int32_t* src = nullptr; // source entry of every residual entry; null: src_base + its position
        int64_t src_base = 0;
        bool want_src = false;        // cfg.value_map: keep the slot map of pb_val
        std::vector<int32_t> own_col, own_src;
        std::vector<double> own_val;
    } deferred;
    bool pb_host_missing = false;

    ehyb_stats stats{};
};

// The host's share of the panel builder (er_panel.cpp), common to the host and the device route
struct PanelGeometry {
    int W = 0;
    std::vector<int32_t> seg_first;    // [segments + 1] column segments
    std::vector<int32_t> seg_panel0;   // [segments + 1] first panel of every segment
    std::vector<int32_t> panel_first;  // [panels + 1] first column of every panel
    std::vector<int32_t> rb_first;     // [row blocks + 1] first row of every row block of pass 2
    std::vector<uint8_t> row_assign;   // [rows] pb_assign only: the row's partition has no window
};
int panel_geometry(const Config& cfg, const HostLayout& L, const int32_t* cnt_row, int64_t nnz_er, PanelGeometry* G);
int64_t panel_pass1_items(const Config& cfg, const PanelGeometry& G, const std::vector<int64_t>& pstart, int64_t nnz_er, HostLayout* L);
void panel_finish(const PanelGeometry& G, const std::vector<int64_t>& rb_count, int64_t staged, int64_t padded, int64_t n_pieces, int64_t n_jumps,
                  HostLayout* L);

// defer_panel: a panel form that is certain (cfg.er_mode = 2, or partitions given up) is left to the device --
// out->deferred.pending, no CSR segments either; the views of out->deferred may point into *m
int build_layout(const matrixCOO* m, int row_begin, int row_end, const Config& cfg, HostLayout* out,
                 const std::vector<uint8_t>* part_to_er = nullptr, int local_lo = -1, int local_hi = -1, bool defer_panel = false,
                 bool stats_only = false);  // stats_only: stop after the windows and slab widths (plan.cpp's sample)
int create_host_plan(const matrixCOO* m, int row_begin, int row_end, const ehyb_config* cfg, int n_col_segs, const int* col_seg_first,
                     bool defer_panel, ehyb_plan** plan);                         // plan.cpp
int spmv_xy_partials(const ehyb_plan* P);            // ehyb_hip.hip: y = A x with x . y on the side -- partials it leaves, 0 = not this plan
int spmv_xy(ehyb_plan* P, const double* x, double* y, void* stream, double* xy_partials);
int build_panel_on_device(ehyb_plan* P);              // er_panel_dev.hip: P->host.deferred -> the d_pb_* arrays
int materialize_panel_host(ehyb_plan* P);             // er_panel_dev.hip: pb_* streams of a device-built plan back to the host
int build_panel_residual(const Config& cfg, HostLayout* L);  // er_panel.cpp; reads the CSR residual of *L
void encode_panel_slots(HostLayout* L);                      // er_panel.cpp; pb_col + pb_dst -> pb_colf, pb_chunk, pb_jump
bool sym_storage_suits(const matrixCOO* m);  // spmvGPuEHYB's own choice of the storage (plan.cpp)
int windows_that_do_not_pay(const HostLayout& H, int pct, std::vector<uint8_t>* to_er, int64_t* entries_moved);  // plan.cpp

// ---------------------------------------------------------------- partitioner
// *by_degree (may be null) = true: the parts are blocks of the degree order (EHYB_PART_DEGREE, or EHYB_PART_AUTO on a
// graph that does not coarsen) -- the caller should number the rows of a part in that order (degree_order)
int partition_graph(int n, const int64_t* xadj, const int* adjncy, const int* vwgt, int nparts,
                    int max_part_w, const Config& cfg, int* part, int64_t* edgecut, bool* by_degree = nullptr);
void degree_order(int n, const int64_t* xadj, std::vector<int>* order);
// the finest-level refinement of partition_graph alone, on a partition that exists: greedy boundary refinement, the hard cap, a polish
// (adjncy may hold self loops: they are skipped)
int refine_partition(int n, const int64_t* xadj, const int* adjncy, const int* vwgt, int nparts, int max_part_w, const Config& cfg, int* part,
                     int64_t* edgecut);

// Optional mt-metis backend: resolved at run time from the process image (weak symbol) --
// see INTEGRATION.md.  Returns false when not linked.
bool mtmetis_available();
int mtmetis_partition(int n, const int64_t* xadj, const int* adjncy, int nparts, int nthreads,
                      int* part, int64_t* edgecut);

// ---------------------------------------------------------------- misc
double wall_seconds();
// cfg.col_map: the host builders look columns up in ONE array over all columns per host thread where that array is small enough to be
// looked into at random cheaply -- 16 MiB, 4 M columns (the bench matrix: 3.7 MB) -- and fall back to sorted lists / hash tables of a
// partition's own candidates beyond (R-MAT 2^24: 64 MiB per thread, first touched and zeroed per thread and per window sample, cost the
// plan 0.36 s, and a partition's 10^5 candidates scattered over it miss the caches and the TLB every time).
inline bool col_map_fits(const Config& cfg, int n_cols) { return cfg.col_map != 2 && (int64_t)n_cols * 4 <= (int64_t(16) << 20); }
void prefault(void* p, size_t bytes);  // fresh pages of an array about to be filled, mapped in one sweep instead of a trap per page
template <class T, class A>
inline void prefault_vector(std::vector<T, A>& v, size_t n)  // reserve + prefault: the resize / push_backs that follow touch mapped pages
{
    v.reserve(n);
    prefault((void*)v.data(), n * sizeof(T));
}
int default_host_threads();  // OpenMP threads when cfg.host_threads is 0: capped by affinity and cgroup quota
// The library's OpenMP regions run on cfg.host_threads (or the default above) threads; the caller's
// own OpenMP setting (the calling thread's nthreads-var) is put back when the entry point returns.
struct OmpScope {
    int saved;
    explicit OmpScope(int want);
    explicit OmpScope(const ehyb_config* cfg) : OmpScope(cfg ? cfg->host_threads : 0) {}
    ~OmpScope();
    OmpScope(const OmpScope&) = delete;
    OmpScope& operator=(const OmpScope&) = delete;
};
inline uint64_t splitmix64(uint64_t& s)
{
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
inline uint64_t mix64(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

}  // namespace ehyb

// The plan object behind the opaque handle.
struct ehyb_plan {
    ehyb::Config cfg;
    ehyb::HostLayout host;
    bool uploaded = false;
    int device = -1;
    std::vector<int32_t> perm;  // reorderList a cached plan was saved with (ehyb_plan_load), else empty
    // device arrays (same names as HostLayout; what the kernels do not read stays on the host)
    int32_t* d_halo_cols = nullptr;
    int32_t* d_segs = nullptr;
    double* d_ell_val = nullptr;
    uint32_t* d_ell_col = nullptr;
    uint8_t* d_lane_group = nullptr;
    uint32_t* d_slab_meta = nullptr;
    int32_t* d_items = nullptr;
    int32_t* d_item_map = nullptr;      // ehyb_plan_tune: item of every ELL workgroup (null: the built-in order)
    std::vector<int32_t> item_map;      // its host copy (a property of the device the plan was tuned on: not saved)
    std::vector<int32_t*> retired_item_maps;  // maps a later ehyb_plan_tune replaced: kept until the plan goes (captured graphs)
    int64_t* d_er_seg_ptr = nullptr;
    int32_t* d_er_seg_row = nullptr;
    int32_t* d_er_col = nullptr;
    double* d_er_val = nullptr;
    int32_t* d_er_blocks = nullptr;
    uint16_t* d_slab_lrow = nullptr;
    double* d_pb_val = nullptr;
    uint16_t* d_pb_colf = nullptr;
    uint32_t* d_pb_chunk = nullptr;
    uint32_t* d_pb_jump = nullptr;
    int32_t* d_pb_units1 = nullptr;
    int32_t* d_pb_items1 = nullptr;
    int* d_pb_queue = nullptr;          // pass 1 work queues: [16 k] items taken of XCD k's eighth, [128] workgroups gone
    uint16_t* d_pb_row = nullptr;
    int32_t* d_pb_units2 = nullptr;
    double* d_pb_partial = nullptr;  // [pb_partials] written by pass 1, read by pass 2: one multiply at a time per plan
    // slot maps on the device, uploaded by the first ehyb_plan_set_values
    int32_t* d_ell_src = nullptr;
    int32_t* d_ell_src2 = nullptr;
    int32_t* d_er_src = nullptr;
    int32_t* d_pb_src = nullptr;
    bool host_values_stale = false;  // the device values were refilled: the host copy no longer matches
    // direction of the next ELL launch's walk (cfg.ell_alternate), and of pass 1 of the panel residual: atomics, so that host
    // threads that multiply with one plan on their own streams (ELL-only plans: include/ehyb.h) each draw a direction
    std::atomic<int> launch_parity{0};
    std::atomic<int> panel_parity{0};
};
