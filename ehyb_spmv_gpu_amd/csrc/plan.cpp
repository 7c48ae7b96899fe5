// Host half of the plan object: layout construction and read-only views (no HIP calls).
#include "ehyb_internal.h"

#include <algorithm>
#include <new>

using namespace ehyb;

namespace ehyb {

// What spmvGPuEHYB does when nobody tells it the storage (cfg == NULL): symmetric pair storage iff
//   * the matrix has at least EHYB_SYM_MIN_ROWS rows (below that plain storage is faster),
//   * it arrives with partitions, enough of them to fill the chip with one workgroup each, and every
//     partition leaves room in the symmetric window for its x image AND its y accumulators -- which is
//     how matrixReorder (symmetric files only, solver_test.c:369-370) sizes them,
//   * a sample of its off-diagonal entries has bitwise equal mirror images (a_ij == a_ji).
// The result never depends on the answer (entries without a partner are stored as they are); only
// the speed does.
// Does the LDS window of a partition pay, next to the panel form of the residual?
//   window:  8 B per stored value (padding included) + its column words, 8 B per staged own row, one
//            cache line (64 B of fabric and an L2 request -- the scarce resource) per
//            gathered halo column;
//   panel:   ~30 B streamed per entry.
// R-MAT 2^22: the ELL launch took 100 us for 9.6 M entries (10.4 ns each: 40 % padding, 3.3 M halo
// gathers) while the panel residual did 23.3 M in 142-155 us (6.4 ns each).  -> number of partitions
// to move, flags per partition of the layout.
int windows_that_do_not_pay(const HostLayout& H, int pct, std::vector<uint8_t>* to_er, int64_t* entries_moved)
{
    const int np = H.n_parts;
    to_er->assign((size_t)np, 0);
    std::vector<int64_t> stored((size_t)np, 0), words((size_t)np, 0);
    for (size_t s = 0; s < H.slab_part.size(); ++s) {
        stored[H.slab_part[s]] += (int64_t)(H.slab_meta[4 * s + 3] >> 16) * 2 * kSlabRows;
        words[H.slab_part[s]] += (int64_t)(H.slab_meta[4 * s + 3] >> 16) * ((H.slab_meta[4 * s + 3] & 0x3F) + 1);
    }
    int count = 0;
    *entries_moved = 0;
    for (int p = 0; p < np; ++p) {
        const int64_t entries = H.part_nnz_ell[p];
        if (entries == 0) continue;
        const int64_t halo = H.halo_ptr[p + 1] - H.halo_ptr[p];
        const int64_t window = 8 * stored[p] + 4 * words[p] + 8 * (int64_t)H.win_len[p] + 64 * halo;
        if (window * 100 > 30 * entries * pct) {  // more than 10 % dearer than the panel form
            (*to_er)[p] = 1;
            ++count;
            *entries_moved += entries;
        }
    }
    return count;
}

// Will every window of this matrix be given up anyway?  The two-build route above costs a power-law matrix its whole
// layout twice (R-MAT 2^24: 7.6 s of 13.7 s).  Decided from a SAMPLE instead: 24 partitions, evenly spread, the first
// one among them (the hubs of a degree-ordered matrix), are laid out on their own -- a partition's window is a function of
// its rows alone -- and judged by the same rule; if the windows that pay would hold less than 20 % of the entries (the
// all-or-nothing line of the full rule is 25 %), the one and only build starts with every partition in the residual.
// Only tried where the entries show no locality (more than one distinct 128-byte line of x per two consecutive
// entries, as the residual's own test in layout.cpp): a mesh never comes here.  A wrong guess costs speed, never
// correctness: any assignment of partitions to the residual multiplies right.
static bool windows_will_not_pay(const matrixCOO* m, int row_begin, int row_end, const Config& cfg)
{
    const int* rp = m->rowIdx;
    const int64_t k0 = rp[row_begin], k1 = rp[row_end], nnz = k1 - k0;
    if (nnz < (1 << 21) || cfg.sym_pairs == 1 || cfg.er_mode == 1 || cfg.ell_prune == 2 || cfg.window_mode != EHYB_WINDOW_HALO) return false;
    if (!m->partBoundary || m->nParts < 8) return false;
    // locality of the entries as stored
    {
        const int64_t win = 1024, nwin = 64;
        int64_t lines = 0;
        std::vector<int32_t> tmp((size_t)win);
        for (int64_t w = 0; w < nwin; ++w) {
            const int64_t at = k0 + (nnz - win) * w / (nwin - 1);
            for (int64_t k = 0; k < win; ++k) tmp[(size_t)k] = m->J[at + k] >> 4;
            std::sort(tmp.begin(), tmp.end());
            lines += std::unique(tmp.begin(), tmp.end()) - tmp.begin();
        }
        if (lines * 2 <= nwin * win) return false;
    }
    std::vector<int> parts;  // partitions inside the plan's rows
    for (int p = 0; p < m->nParts; ++p)
        if (m->partBoundary[p] >= row_begin && m->partBoundary[p + 1] <= row_end && m->partBoundary[p + 1] > m->partBoundary[p]) parts.push_back(p);
    if (parts.size() < 8) return false;
    // the first partition (the hubs of a degree-ordered matrix: it can hold a tenth of all entries) and 23 more, evenly spread
    std::vector<int> pick;
    for (int i = 0; i < 24; ++i) pick.push_back(parts[(parts.size() - 1) * (size_t)i / 23]);
    pick.erase(std::unique(pick.begin(), pick.end()), pick.end());
    Config quiet = cfg;
    quiet.verbose = 0;
    quiet.er_mode = 1;     // no panel form for the samples
    quiet.value_map = 0;
    quiet.direct = 2;
    int64_t seen = 0, kept_first = 0, kept_rest = 0;
    // the samples are laid out side by side (a one-partition layout runs on one thread; the hub partition -- first in the
    // list, a fifth of a degree-ordered R-MAT -- takes as long as the 23 others together)
    std::vector<int64_t> s_nnz(pick.size(), 0), s_kept(pick.size(), 0);
    bool failed = false;
    quiet.host_threads = 1;
    {
        OmpScope omp_scope(cfg.host_threads);
#pragma omp parallel for schedule(dynamic, 1)
        for (size_t i = 0; i < pick.size(); ++i) {
            const int p = pick[i];
            HostLayout S;
            int rc;
            try {
                rc = build_layout(m, m->partBoundary[p], m->partBoundary[p + 1], quiet, &S, nullptr, row_begin, row_end, false, true);
            } catch (const std::bad_alloc&) {
                rc = EHYB_ERR_ALLOC;
            }
            if (rc != EHYB_OK) {
#pragma omp atomic write
                failed = true;
                continue;
            }
            std::vector<uint8_t> to_er;
            int64_t moved = 0;
            windows_that_do_not_pay(S, cfg.prune_pct, &to_er, &moved);
            s_nnz[i] = S.stats.nnz;
            s_kept[i] = S.stats.nnz_ell - moved;
        }
    }
    if (failed) {
        clear_error();
        return false;
    }
    for (size_t i = 0; i < pick.size(); ++i) {
        seen += s_nnz[i];
        (i == 0 ? kept_first : kept_rest) += s_kept[i];
    }
    // the first partition counts for itself, the others for their share of the rest
    const double kept_est = (double)kept_first + (double)kept_rest * (double)(parts.size() - 1) / (double)std::max<size_t>(1, pick.size() - 1);
    if (cfg.verbose)
        printf("window sample: %zu partitions laid out (%lld entries): about %.1f %% of the %lld entries would sit in windows that pay\n", pick.size(),
               (long long)seen, 100.0 * kept_est / (double)nnz, (long long)nnz);
    return kept_est < 0.20 * (double)nnz;  // the full rule gives every window up below 25 %
}

bool sym_storage_suits(const matrixCOO* m)
{
    const int n = m->dimension;
    if (n < EHYB_SYM_MIN_ROWS || !m->partBoundary || !m->rowIdx || !m->J || !m->V) return false;
    const int np = m->nParts;
    if (np < kNumCU / 2 || m->partBoundary[0] != 0 || m->partBoundary[np] != n) return false;
    ehyb_config zs;
    memset(&zs, 0, sizeof zs);
    zs.sym_pairs = 1;
    const Config cs = resolve_config(&zs);
    int max_rows = 0;
    for (int p = 0; p < np; ++p) {
        if (m->partBoundary[p + 1] < m->partBoundary[p]) return false;
        max_rows = std::max(max_rows, m->partBoundary[p + 1] - m->partBoundary[p]);
    }
    if (2 * (max_rows + 1) > cs.lds_doubles - 2) return false;
    // symmetry probe: up to 4096 entries, evenly spread
    const int64_t nnz = m->totalNum;
    if (nnz <= 0) return false;
    const int64_t step = std::max<int64_t>(1, nnz / 4096);
    int64_t seen = 0, paired = 0;
    int row = 0;
    for (int64_t k = 0; k < nnz; k += step) {
        while (row + 1 < n && m->rowIdx[row + 1] <= k) ++row;
        const int j = m->J[k];
        if (j == row || (unsigned)j >= (unsigned)n) continue;
        ++seen;
        for (int q = m->rowIdx[j]; q < m->rowIdx[j + 1]; ++q)
            if (m->J[q] == row && m->V[q] == m->V[k]) {
                ++paired;
                break;
            }
    }
    return seen > 0 && paired * 10 >= seen * 9;
}

}  // namespace ehyb

extern "C" {

int ehyb_plan_create_host(const matrixCOO* m, int row_begin, int row_end, const ehyb_config* cfg,
                          ehyb_plan** plan)
{
    return ehyb_plan_create_host_segs(m, row_begin, row_end, cfg, 0, nullptr, plan);
}

int ehyb_plan_create_host_segs(const matrixCOO* m, int row_begin, int row_end, const ehyb_config* cfg,
                               int n_col_segs, const int* col_seg_first, ehyb_plan** plan)
{
    return ehyb::create_host_plan(m, row_begin, row_end, cfg, n_col_segs, col_seg_first, false, plan);
}

}  // extern "C"

// defer_panel (ehyb_plan_create[_segs] only): a panel form that is certain is left to the device -- the plan comes back with
// host.deferred.pending, whose views may point into *m: the caller finishes it (build_panel_on_device) before m can change
int ehyb::create_host_plan(const matrixCOO* m, int row_begin, int row_end, const ehyb_config* cfg, int n_col_segs, const int* col_seg_first,
                           bool defer_panel, ehyb_plan** plan)
{
    clear_error();
    if (!plan) EHYB_FAIL(EHYB_ERR_ARG, "ehyb_plan_create_host: null output");
    *plan = nullptr;
    if (!m) EHYB_FAIL(EHYB_ERR_ARG, "ehyb_plan_create_host: null matrix");
    if (n_col_segs < 0 || (n_col_segs > 0 && !col_seg_first)) EHYB_FAIL(EHYB_ERR_ARG, "ehyb_plan_create_host_segs: bad column segments");
    for (int s = 0; s < n_col_segs; ++s)
        if (col_seg_first[s + 1] < col_seg_first[s] || (s > 0 && (col_seg_first[s] & 1)))
            EHYB_FAIL(EHYB_ERR_ARG, "ehyb_plan_create_host_segs: column segment %d starts at %d (segments ascend and start on even columns)", s, col_seg_first[s]);
    if (n_col_segs > 0 && (col_seg_first[0] != 0 || col_seg_first[n_col_segs] != m->dimension))
        EHYB_FAIL(EHYB_ERR_ARG, "ehyb_plan_create_host_segs: the column segments must span [0, %d)", m->dimension);
    ehyb_plan* P = new (std::nothrow) ehyb_plan();
    if (!P) EHYB_FAIL(EHYB_ERR_ALLOC, "ehyb_plan_create_host: out of memory");
    P->cfg = resolve_config(cfg);
    if (n_col_segs > 0) P->host.col_seg_first.assign(col_seg_first, col_seg_first + n_col_segs + 1);
    int rc;
    try {
        bool decided = false;
        if (windows_will_not_pay(m, row_begin, row_end, P->cfg)) {
            // straight to the layout the two-build route would end with: every partition in the (panel-form) residual
            std::vector<uint8_t> all(m->nParts > 0 ? (size_t)m->nParts + 64 : 64, 1);
            HostLayout direct_to;
            direct_to.col_seg_first = P->host.col_seg_first;
            // (the layout's own partition list may be longer than the caller's -- partitions cut down to the window --
            // so the flags cover any index: every partition goes)
            all.assign((size_t)(row_end - row_begin) / kSlabRows + (size_t)m->nParts + 64, 1);
            const int rc0 = build_layout(m, row_begin, row_end, P->cfg, &direct_to, &all, -1, -1, defer_panel);
            if (rc0 == EHYB_OK && (direct_to.er_panel || direct_to.deferred.pending)) {
                P->host = std::move(direct_to);
                decided = true;
            } else {
                clear_error();
            }
        }
        // (the first build of the two-build route needs no panel form to judge the windows: deferred as well when certain)
        rc = decided ? EHYB_OK : build_layout(m, row_begin, row_end, P->cfg, &P->host, nullptr, -1, -1, defer_panel);
        // Where the residual runs in panel form (a large residual without locality: R-MAT), a partition
        // whose window does not pay is better off in the residual whole: built a second time with those.
        if (!decided && rc == EHYB_OK && (P->host.er_panel || P->host.deferred.pending) && !P->host.sym && P->cfg.er_mode != 1 && P->cfg.ell_prune != 2) {
            std::vector<uint8_t> to_er;
            int64_t moved = 0;
            if (windows_that_do_not_pay(P->host, P->cfg.prune_pct, &to_er, &moved) > 0) {
                // The windows that are left must carry the ELL launch: its staging of 160 KiB windows with thousands
                // of gathered halo columns, for a few partitions on a few CUs, is a fixed cost.  Measured on R-MAT
                // (tools/panel_sweep.py --prune-pct): 2^22 with 18 % of the entries left in windows 216 us, with none
                // 194 us; 2^24 with 8 % left 859 us, with none 859 us; partial pruning in between was never better
                // than either end.  Below a quarter of the entries the rest goes to the panel form as well.
                const int64_t kept = P->host.stats.nnz_ell - moved;
                if (kept * 4 < P->host.stats.nnz) {
                    for (int p = 0; p < P->host.n_parts; ++p)
                        if (P->host.part_nnz_ell[p] > 0) to_er[(size_t)p] = 1;
                    moved = P->host.stats.nnz_ell;
                }
                if (P->cfg.verbose) printf("%lld ELL entries sit in windows that cost more than the panel residual: rebuilt with those partitions in the residual\n", (long long)moved);
                // A failure of the rebuild (memory: both layouts are alive here; the 32-bit limits of the panel builder once
                // every ELL entry has moved) is not a failure of the plan: the first layout is complete and valid.
                HostLayout again;
                again.col_seg_first = P->host.col_seg_first;
                int rc2;
                try {
                    rc2 = build_layout(m, row_begin, row_end, P->cfg, &again, &to_er, -1, -1, defer_panel);
                } catch (const std::bad_alloc&) {
                    rc2 = EHYB_ERR_ALLOC;
                }
                if (rc2 == EHYB_OK && (again.er_panel || again.deferred.pending))
                    P->host = std::move(again);
                else if (P->cfg.verbose)
                    printf("rebuild without those windows failed (%d): keeping the first layout\n", rc2);
                clear_error();
            }
        }
    } catch (const std::bad_alloc&) {
        set_error("ehyb_plan_create_host: out of memory while building the layout");
        rc = EHYB_ERR_ALLOC;
    }
    if (rc != EHYB_OK) {
        delete P;
        return rc;
    }
    *plan = P;
    return EHYB_OK;
}

extern "C" {

int ehyb_plan_stats(const ehyb_plan* plan, ehyb_stats* out)
{
    if (!plan || !out) EHYB_FAIL(EHYB_ERR_ARG, "ehyb_plan_stats: null argument");
    *out = plan->host.stats;
    return EHYB_OK;
}

int ehyb_plan_host_array(const ehyb_plan* plan, int which, const void** ptr, int64_t* count)
{
    if (!plan || !ptr || !count) EHYB_FAIL(EHYB_ERR_ARG, "ehyb_plan_host_array: null argument");
    // a plan whose panel form was built on the device fetches those streams the first time anybody asks (the plan is
    // logically unchanged: const for the caller)
    if (plan->host.pb_host_missing && which >= EHYB_ARR_PB_VAL) {
        const int rc = materialize_panel_host(const_cast<ehyb_plan*>(plan));
        if (rc != EHYB_OK) return rc;
    }
    const HostLayout& H = plan->host;
#define VIEW(v)                      \
    *ptr = (const void*)(v).data(); \
    *count = (int64_t)(v).size();   \
    return EHYB_OK
    switch (which) {
        case EHYB_ARR_PART_BOUNDARY: VIEW(H.part_boundary);
        case EHYB_ARR_WIN_LEN: VIEW(H.win_len);
        case EHYB_ARR_HALO_PTR: VIEW(H.halo_ptr);
        case EHYB_ARR_HALO_COLS: VIEW(H.halo_cols);
        case EHYB_ARR_SLAB_PAIR_PTR: VIEW(H.slab_pair_ptr);
        case EHYB_ARR_SLAB_ROW: VIEW(H.slab_row);
        case EHYB_ARR_SLAB_PART: VIEW(H.slab_part);
        case EHYB_ARR_ELL_VAL: VIEW(H.ell_val);
        case EHYB_ARR_ELL_COL: VIEW(H.ell_col);
        case EHYB_ARR_ITEMS: VIEW(H.items);
        case EHYB_ARR_ER_SEG_PTR: VIEW(H.er_seg_ptr);
        case EHYB_ARR_ER_SEG_ROW: VIEW(H.er_seg_row);
        case EHYB_ARR_ER_COL: VIEW(H.er_col);
        case EHYB_ARR_ER_VAL: VIEW(H.er_val);
        case EHYB_ARR_SLAB_COL_PTR: VIEW(H.slab_col_ptr);
        case EHYB_ARR_LANE_GROUP: VIEW(H.lane_group);
        case EHYB_ARR_SLAB_META: VIEW(H.slab_meta);
        case EHYB_ARR_SEGS: VIEW(H.segs);
        case EHYB_ARR_PERM: VIEW(plan->perm);
        case EHYB_ARR_SLAB_LROW: VIEW(H.slab_lrow);
        case EHYB_ARR_PB_VAL: VIEW(H.pb_val);
        case EHYB_ARR_PB_COL: VIEW(H.pb_col);
        case EHYB_ARR_PB_DST: VIEW(H.pb_dst);
        case EHYB_ARR_PB_UNITS1: VIEW(H.pb_units1);
        case EHYB_ARR_PB_ROW: VIEW(H.pb_row);
        case EHYB_ARR_PB_UNITS2: VIEW(H.pb_units2);
        case EHYB_ARR_PB_COLF: VIEW(H.pb_colf);
        case EHYB_ARR_PB_CHUNK: VIEW(H.pb_chunk);
        case EHYB_ARR_PB_JUMP: VIEW(H.pb_jump);
        case EHYB_ARR_ELL_SRC: VIEW(H.ell_src);
        case EHYB_ARR_ER_SRC: VIEW(H.er_src);
        case EHYB_ARR_PB_SRC: VIEW(H.pb_src);
        case EHYB_ARR_ELL_SRC2: VIEW(H.ell_src2);
        case EHYB_ARR_COL_SEG_FIRST: VIEW(H.col_seg_first);
        case EHYB_ARR_PB_SEG_ITEM: VIEW(H.pb_seg_item);
        case EHYB_ARR_PB_ITEMS1: VIEW(H.pb_items1);
        case EHYB_ARR_ER_BINS:
            *ptr = (const void*)H.er_bins;
            *count = 8;
            return EHYB_OK;
        default: break;
    }
#undef VIEW
    EHYB_FAIL(EHYB_ERR_ARG, "ehyb_plan_host_array: unknown array id %d", which);
}

}  // extern "C"
