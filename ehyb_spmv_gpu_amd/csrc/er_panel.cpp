// Panel form of a LARGE residual WITHOUT locality: the random gather of x is traded for two streaming passes.
//
// The reference multiplies its residual ("ER") rows straight from global memory: value and column
// streamed, x[column] gathered wherever it lies (kernel.cu:169-194).  On a matrix without locality
// (R-MAT) every such gather is its own request to L2 -- 65 % of them hit, but 25 M requests per launch
// are what bounds the CSR residual kernel of this library (ehyb_er_kernel: 210 us for the 23 M residual
// entries of R-MAT 2^22, ~110 G entries/s at any size).  The explicit cache of the ELL
// part (x window in LDS) removes exactly that cost, but only for columns near the row's own partition.
// This form extends the idea to ALL columns:
//
//   pass 1 "scale"   entries grouped by COLUMN PANEL (er_panel_cols consecutive columns, default 8192 =
//                    64 KiB).  A workgroup stages its panel of x in LDS (coalesced) and streams (value,
//                    16-bit local column, slot): product = value * x_lds[column]; products of one row
//                    that sit next to each other inside a 64-entry chunk are summed in the wave's LDS
//                    words and ONE partial per such piece is written to partial[slot] -- a hub row of
//                    50,000 entries leaves a few hundred partials.
//   pass 2 "reduce"  partials grouped by ROW BLOCK (consecutive rows holding about equally many
//                    entries, at most er_block_rows rows, default 2048 = 16 KiB).  A workgroup keeps the
//                    block's y accumulators in LDS, streams (partial, 16-bit local row), adds
//                    (ds_add_f64) and finally y[row] += accumulator.
//
// Inside a panel the entries are ordered by row, and the partial slots are numbered by (row block,
// panel, row): a pass-1 workgroup writes runs of consecutive slots, a pass-2 workgroup reads ONE
// contiguous range.  Every byte of both passes is streamed: 14 B read + <= 8 B written + <= 10 B read
// per entry instead of 12 B + an L2 request.  Rows of any length and any skew are fine; no global
// atomics.  Chosen by build_layout for residuals of 2^21 entries and more whose columns show no
// locality (or cfg.er_mode = 2); measured 1.35-1.4x the CSR form on R-MAT 2^22 / 2^24, behind it
// where neighbouring rows read neighbouring columns (kkt3d on contiguous partitions).
#include "ehyb_internal.h"

#include <omp.h>

#include <parallel/algorithm>

#include <algorithm>
#include <numeric>

namespace ehyb {

// What pass 1 really streams instead of the 4-byte slot of every entry (pb_dst stays on the host as the
// definition the layout tests read).  Inside a 64-entry chunk the entries are sorted by row, so their
// slots are runs: a piece keeps its slot, the next piece of the same row block has the next slot, and only where
// the chunk crosses into another row block (or into the padding at the end of a panel) the slot jumps.  Hence
//   pb_colf  [entry]  bits 0-13 the panel-local column, bit 15 = first entry of a piece ("head"; lane 0 always),
//                     bit 14 = head whose slot does not follow the previous piece's ("jump"; lane 0 always);
//   pb_chunk [chunk]  index of the chunk's first jump in pb_jump (+ one end entry);
//   pb_jump  [jumps]  slot of the jump piece minus the number of pieces before it in its chunk, so that
//                     slot(lane) = pb_jump[first + jumps up to the lane - 1] + pieces before the lane's  (mod 2^32)
//                     for every lane, whichever jump it follows; the padding piece at the end of a panel gets
//                     the value that makes this 0xFFFFFFFF = nothing is stored.
// R-MAT 2^24: 4 B per entry become ~0.6 B.
void encode_panel_slots(HostLayout* L)
{
    const int64_t padded = (int64_t)L->pb_dst.size();
    const int64_t chunks = padded / 64;
    L->pb_colf.assign((size_t)padded, 0);
    L->pb_chunk.assign((size_t)chunks + 1, 0);
    L->pb_jump.clear();
    if (padded == 0) return;
    const uint32_t* dst = L->pb_dst.data();
    const uint16_t* col = L->pb_col.data();
    // jumps per chunk first (parallel), then their places
    std::vector<int64_t> jfirst((size_t)chunks + 1, 0);
#pragma omp parallel for schedule(static, 256)
    for (int64_t c = 0; c < chunks; ++c) {
        const uint32_t* d = dst + c * 64;
        int j = 1;
        uint32_t prev_head = d[0];
        for (int l = 1; l < 64; ++l)
            if (d[l] != d[l - 1]) {
                j += d[l] != prev_head + 1;
                prev_head = d[l];
            }
        jfirst[(size_t)c + 1] = j;
    }
    for (int64_t c = 0; c < chunks; ++c) jfirst[(size_t)c + 1] += jfirst[(size_t)c];
    L->pb_jump.resize((size_t)jfirst[(size_t)chunks]);
    L->pb_chunk[(size_t)chunks] = (uint32_t)jfirst[(size_t)chunks];
#pragma omp parallel for schedule(static, 256)
    for (int64_t c = 0; c < chunks; ++c) {
        const uint32_t* d = dst + c * 64;
        uint16_t* f = L->pb_colf.data() + c * 64;
        int64_t j = jfirst[(size_t)c];
        L->pb_chunk[(size_t)c] = (uint32_t)j;
        uint32_t prev_head = d[0], pieces = 0;  // pieces begun before the current lane's
        f[0] = (uint16_t)(col[c * 64] | 0xC000u);
        L->pb_jump[(size_t)j++] = d[0];
        for (int l = 1; l < 64; ++l) {
            uint16_t w = col[c * 64 + l];
            if (d[l] != d[l - 1]) {
                ++pieces;
                w |= 0x8000u;
                if (d[l] != prev_head + 1) {
                    w |= 0x4000u;
                    L->pb_jump[(size_t)j++] = d[l] - pieces;  // mod 2^32: 0xFFFFFFFF - pieces for the padding piece
                }
                prev_head = d[l];
            }
            f[l] = w;
        }
    }
}


// ---- the parts of the builder that stay on the host whichever side deals the entries out (the device side is
// er_panel_dev.hip): row blocks and panels before the entries are looked at, work items and units after they are counted

int panel_geometry(const Config& cfg, const HostLayout& L, const int32_t* cnt_row, int64_t nnz_er, PanelGeometry* G)
{
    const int n_cols = L.n_cols;
    const int W = cfg.er_panel_cols;
    const int rows_max = cfg.er_block_rows;
    G->W = W;
    if (nnz_er + 64 * ((int64_t)n_cols / W + 2 + (int64_t)L.col_seg_first.size()) >= 0x7FFFFF00ll)
        EHYB_FAIL(EHYB_ERR_ARG, "build_panel_residual: residual of %lld entries too large for 32-bit offsets", (long long)nnz_er);
    // ---- row blocks: consecutive rows with about `target` entries each, at most rows_max rows
    const int row_begin = L.row_begin, nrows = L.row_end - L.row_begin;
    // (cfg.er_units1 / er_units2 = work units aimed at per pass: tools/panel_sweep.py)
    const int64_t target = std::min<int64_t>(std::max<int64_t>(nnz_er / cfg.er_units2, 4096), 1 << 20);
    std::vector<int32_t>& rb_first = G->rb_first;  // first row (plan numbering) of every block, + end
    rb_first.clear();
    // pb_assign: rows of partitions without a window get y from pass 2 alone, so a block never mixes them with
    // rows the ELL launch writes
    std::vector<uint8_t>& row_assign = G->row_assign;
    row_assign.assign(L.pb_assign ? (size_t)nrows : 0, 0);
    if (L.pb_assign)
        for (int p = 0; p < L.n_parts; ++p)
            if (L.part_windowless[p])
                std::fill(row_assign.begin() + (L.part_boundary[p] - row_begin), row_assign.begin() + (L.part_boundary[p + 1] - row_begin), (uint8_t)1);
    {
        int64_t acc = 0;
        int first = 0;
        rb_first.push_back(row_begin);
        for (int r = 0; r < nrows; ++r) {
            // close the block in front of this row if taking it would overshoot
            // (cfg.row_split: the rows from it on are closed by a pass 2 of their own, ehyb_spmv_part / EHYB_PART_LAST_FOREIGN)
            if (r > first && (acc + cnt_row[r] > target || r - first >= rows_max || (L.pb_assign && row_assign[r] != row_assign[r - 1]) ||
                              (cfg.row_split > 0 && row_begin + r == cfg.row_split))) {
                rb_first.push_back(row_begin + r);
                first = r;
                acc = 0;
            }
            acc += cnt_row[r];
        }
        rb_first.push_back(row_begin + nrows);
    }
    // ---- panels: W columns each, starting afresh at every column segment (one segment = the whole matrix unless the
    // caller named more: multi-GPU, the x entries of a segment arrive together)
    std::vector<int32_t>& seg_first = G->seg_first;
    seg_first = L.col_seg_first;
    if (seg_first.size() < 2) seg_first = {0, n_cols};
    const int n_segs = (int)seg_first.size() - 1;
    if (seg_first[0] != 0 || seg_first[n_segs] != n_cols) EHYB_FAIL(EHYB_ERR_ARG, "build_panel_residual: column segments must span [0, %d)", n_cols);
    G->seg_panel0.assign((size_t)n_segs + 1, 0);
    G->panel_first.clear();
    for (int s = 0; s < n_segs; ++s) {
        if (seg_first[s + 1] < seg_first[s] || (s > 0 && (seg_first[s] & 1)))
            EHYB_FAIL(EHYB_ERR_ARG, "build_panel_residual: column segment %d starts at %d (segments ascend and start on even columns)", s, seg_first[s]);
        for (int c = seg_first[s]; c < seg_first[s + 1]; c += W) G->panel_first.push_back(c);
        G->seg_panel0[s + 1] = (int32_t)G->panel_first.size();
    }
    G->panel_first.push_back(n_cols);
    return EHYB_OK;
}

// ---- work of pass 1: ITEMS of equal cost, one workgroup each; an item is a run of UNITS {first column, columns,
// first entry, end entry} -- a stretch of one panel's entries, multiples of 64 -- and stages a panel once per unit.
// (Round 2: one workgroup per unit of padded/2048 entries, handed out by the hardware.  A heavy panel -- the hub
// columns of a degree-ordered matrix hold a third of all entries -- was then staged hundreds of times, and a
// workgroup lived for four loop trips per wave: R-MAT 2^22 moved 139 MB of staging for 350 MB of entries.)
// Cost of a unit = 11 B per entry (value, column word, its share of the chunk records) + 8 B per staged column +
// a fixed 16 KiB for the two barriers and the pipeline fill.  Items never straddle a column segment.
// pstart = padded position of every panel's first entry (+ end).  -> columns staged per multiply
int64_t panel_pass1_items(const Config& cfg, const PanelGeometry& G, const std::vector<int64_t>& pstart, int64_t nnz_er, HostLayout* L)
{
    const int W = G.W;
    const std::vector<int32_t>&seg_first = G.seg_first, &seg_panel0 = G.seg_panel0, &panel_first = G.panel_first;
    const int n_segs = (int)seg_first.size() - 1;
    const int64_t kPerEntry = 11, kPerCol = 8, kFixed = 16384;
    L->pb_units1.clear();
    L->pb_items1.clear();
    L->pb_seg_item.assign((size_t)n_segs + 1, 0);
    int64_t staged = 0;
    int64_t total_cost = 0;
    std::vector<int64_t> seg_cost((size_t)n_segs, 0);
    for (int s = 0; s < n_segs; ++s)
        for (int p = seg_panel0[s]; p < seg_panel0[s + 1]; ++p)
            if (pstart[p + 1] > pstart[p]) {
                const int64_t c = kPerEntry * (pstart[p + 1] - pstart[p]) + kPerCol * std::min(W, seg_first[s + 1] - panel_first[p]) + kFixed;
                seg_cost[(size_t)s] += c;
                total_cost += c;
            }
    for (int s = 0; s < n_segs; ++s) {
        if (seg_cost[(size_t)s] == 0) {
            L->pb_seg_item[(size_t)s + 1] = (int32_t)(L->pb_items1.size() / 2);
            continue;
        }
        // this segment's share of the items; the target leaves room for the extra stagings the cuts add
        // items aimed at (cfg.er_units1 = 0): one per 48 k entries, between 512 (two rounds of one workgroup per CU) and
        // 4096 -- R-MAT 2^22 (33 M entries): 1024 items 149 us, 2048 items 166 us (each item stages its 128 KiB panel
        // again); 2^24 (133 M entries): 2048 items 544 us, 4096 items 563 us, 1024 items 663 us
        const int64_t aim = cfg.er_units1 > 0 ? cfg.er_units1 : std::min<int64_t>(4096, std::max<int64_t>(512, nnz_er / 49152));
        const int64_t want = std::max<int64_t>(1, (int64_t)((double)aim * (double)seg_cost[(size_t)s] / (double)total_cost + 0.5));
        const int64_t target = seg_cost[(size_t)s] / want + (kPerCol * W + kFixed) / 2 + 1;
        int64_t item_cost = 0;
        int32_t item_first = (int32_t)(L->pb_units1.size() / 4);
        auto close_item = [&]() {
            const int32_t end = (int32_t)(L->pb_units1.size() / 4);
            if (end > item_first) {
                L->pb_items1.push_back(item_first);
                L->pb_items1.push_back(end);
            }
            item_first = end;
            item_cost = 0;
        };
        for (int p = seg_panel0[s]; p < seg_panel0[s + 1]; ++p) {
            const int32_t cols = std::min(W, seg_first[s + 1] - panel_first[p]);
            const int64_t stage = kPerCol * cols + kFixed;
            int64_t pos = pstart[p];
            while (pos < pstart[p + 1]) {
                // not worth staging a panel for less than a quarter of what it costs to stage it
                if (item_cost > 0 && target - item_cost < stage + stage / 4) close_item();
                const int64_t fit = std::max<int64_t>(64, (target - item_cost - stage) / kPerEntry / 64 * 64);
                const int64_t take = std::min(pstart[p + 1] - pos, fit);
                const int32_t u[4] = {panel_first[p], cols, (int32_t)pos, (int32_t)(pos + take)};
                L->pb_units1.insert(L->pb_units1.end(), u, u + 4);
                staged += cols;
                item_cost += stage + kPerEntry * take;
                pos += take;
                if (item_cost >= target) close_item();
            }
        }
        close_item();
        L->pb_seg_item[(size_t)s + 1] = (int32_t)(L->pb_items1.size() / 2);
    }
    return staged;
}

// pass 2: {first slot, end slot, first row, rows}; blocks without partials are skipped -- unless the block
// ASSIGNS y (rows stored negative): then it is the only writer of its rows.  rb_count = first slot of every row
// block (+ end).  Also the scalars of the form and the bytes its two launches move.
void panel_finish(const PanelGeometry& G, const std::vector<int64_t>& rb_count, int64_t staged, int64_t padded, int64_t n_pieces, int64_t n_jumps,
                  HostLayout* L)
{
    const std::vector<int32_t>& rb_first = G.rb_first;
    const int n_rb = (int)rb_first.size() - 1;
    const int row_begin = L->row_begin;
    L->pb_units2.clear();
    int max_rows = 0;
    int64_t rows_assigned = 0;
    for (int b = 0; b < n_rb; ++b) {
        const bool assign = L->pb_assign && G.row_assign[(size_t)(rb_first[b] - row_begin)] != 0;
        if (rb_count[b + 1] == rb_count[b] && !assign) continue;
        const int rows = rb_first[b + 1] - rb_first[b];
        const int32_t u[4] = {(int32_t)rb_count[b], (int32_t)rb_count[b + 1], rb_first[b], assign ? -rows : rows};
        L->pb_units2.insert(L->pb_units2.end(), u, u + 4);
        max_rows = std::max(max_rows, rows);
        rows_assigned += assign ? rows : 0;
    }
    L->pb_panel_cols = G.W;
    L->pb_rows_max = max_rows;
    L->pb_partials = n_pieces;
    L->pb_padded = padded;
    L->er_panel = true;
    // bytes the two launches move: entries (value, column + flags) + chunk records + jump list instead of a 4-byte slot
    // per entry + the staged panels + partials out; partials (value, row) in + the touched y rows read and written
    L->pb_bytes = 10 * padded + 4 * (padded / 64 + 1) + 4 * n_jumps + 8 * staged + 8 * n_pieces + 10 * n_pieces +
                  (L->pb_assign ? 8 * rows_assigned + 16 * std::max<int64_t>(0, L->stats.rows_er - rows_assigned) : 16 * L->stats.rows_er) +
                  16 * (int64_t)(L->pb_units1.size() / 4 + L->pb_units2.size() / 4) + 8 * (int64_t)(L->pb_items1.size() / 2);
}

int build_panel_residual(const Config& cfg, HostLayout* L)
{
    const int64_t nnz_er = (int64_t)L->er_col.size();
    const int64_t nseg = (int64_t)L->er_seg_row.size();
    const int n_cols = L->n_cols;
    const int W = cfg.er_panel_cols;
    L->er_panel = false;
    if (nnz_er == 0) return EHYB_OK;
    double t_lap = wall_seconds();
    auto lap = [&](const char* what) {
        const double now = wall_seconds();
        if (cfg.verbose > 1) printf("  panel form: %-32s %7.1f ms\n", what, (now - t_lap) * 1e3);
        t_lap = now;
    };
    const int32_t* ecol = L->er_col.data();
    const double* evalv = L->er_val.data();

    // ---- row blocks: consecutive rows with about `target` entries each, at most rows_max rows
    const int row_begin = L->row_begin, nrows = L->row_end - L->row_begin;
    std::vector<int32_t> cnt_row((size_t)nrows, 0);
    {
        bool bad_row = false;
#pragma omp parallel for schedule(static, 4096) reduction(|| : bad_row)
        for (int64_t s = 0; s < nseg; ++s) {
            const int r = (L->er_seg_row[s] & 0x7FFFFFFF) - row_begin;
            if ((unsigned)r >= (unsigned)nrows) {
                bad_row = true;
                continue;
            }
            const int32_t len = (int32_t)(L->er_seg_ptr[s + 1] - L->er_seg_ptr[s]);
#pragma omp atomic
            cnt_row[r] += len;  // (a long row comes in several segments)
        }
        if (bad_row) EHYB_FAIL(EHYB_ERR_INTERNAL, "build_panel_residual: residual row outside the plan's rows");
    }
    PanelGeometry G;
    {
        const int rc = panel_geometry(cfg, *L, cnt_row.data(), nnz_er, &G);
        if (rc != EHYB_OK) return rc;
    }
    const std::vector<int32_t>&rb_first = G.rb_first, &seg_first = G.seg_first, &seg_panel0 = G.seg_panel0, &panel_first = G.panel_first;
    const int n_rb = (int)rb_first.size() - 1;
    std::vector<int32_t> rb_of_row((size_t)nrows);
#pragma omp parallel for schedule(dynamic, 64)
    for (int b = 0; b < n_rb; ++b) std::fill(rb_of_row.begin() + (rb_first[b] - row_begin), rb_of_row.begin() + (rb_first[b + 1] - row_begin), b);
    const int n_panels = (int)panel_first.size() - 1;
    lap("row blocks and panels");
    auto panel_of = [&](int c) {
        int s = 0;
        while (c >= seg_first[s + 1]) ++s;  // a handful of segments
        return seg_panel0[s] + (c - seg_first[s]) / W;
    };
    // ---- pass-1 order: by panel, inside a panel by (row, column).  No comparison sort of the entries (round 2 sorted
    // every panel by itself: the hub panel of a degree-ordered R-MAT holds a third of all entries, one thread sorted it
    // for seconds): the segments are taken in ROW order -- a sort of the segments, not of the entries -- cut into one
    // run per thread, and every run deals its entries out to the panels, behind what the runs before it put there (a
    // parallel counting sort by panel that keeps the row order).  A row's entries are dealt in COLUMN order (ties: in
    // stored order): as they lie where the row is stored that way, through a sorted index list where it is not (the
    // reorder step keeps a row's old order, and the segments of a long row are not stored in the row's own order) --
    // the same order the device route's radix sort gives (er_panel_dev.hip).
    std::vector<uint64_t> seg_key((size_t)nseg);
#pragma omp parallel for schedule(static, 4096)
    for (int64_t s = 0; s < nseg; ++s) seg_key[(size_t)s] = (uint64_t)(uint32_t)(L->er_seg_row[s] & 0x7FFFFFFF) << 32 | (uint64_t)s;
    __gnu_parallel::sort(seg_key.begin(), seg_key.end());
    lap("segments in row order");
    const int T = std::max(1, omp_get_max_threads());
    std::vector<int64_t> run_first((size_t)T + 1, nseg);  // first segment (in key order) of every run: about equal entries each
    {
        std::vector<int64_t> ent_before((size_t)nseg + 1, 0);
        for (int64_t i = 0; i < nseg; ++i) {
            const int64_t s = (int64_t)(seg_key[(size_t)i] & 0xFFFFFFFFull);
            ent_before[(size_t)i + 1] = ent_before[(size_t)i] + (L->er_seg_ptr[s + 1] - L->er_seg_ptr[s]);
        }
        for (int t = 0; t <= T; ++t)
            run_first[(size_t)t] = std::lower_bound(ent_before.begin(), ent_before.end(), nnz_er * t / T) - ent_before.begin();
        run_first[0] = 0;
        run_first[(size_t)T] = nseg;
        // a run starts with a row's first segment: the rows are dealt as wholes
        for (int t = 1; t < T; ++t) {
            int64_t& f = run_first[(size_t)t];
            f = std::max(f, run_first[(size_t)t - 1]);
            while (f > 0 && f < nseg && (seg_key[(size_t)f] >> 32) == (seg_key[(size_t)f - 1] >> 32)) ++f;
        }
    }
    std::vector<int64_t> pcount((size_t)n_panels + 1, 0);
    std::vector<int64_t> run_cnt((size_t)T * (size_t)n_panels, 0);  // [run][panel]
    bool bad_col = false;
#pragma omp parallel for schedule(static, 1) reduction(|| : bad_col)
    for (int t = 0; t < T; ++t) {
        int64_t* c = &run_cnt[(size_t)t * (size_t)n_panels];
        for (int64_t i = run_first[(size_t)t]; i < run_first[(size_t)t + 1]; ++i) {
            const int64_t s = (int64_t)(seg_key[(size_t)i] & 0xFFFFFFFFull);
            for (int64_t k = L->er_seg_ptr[s]; k < L->er_seg_ptr[s + 1]; ++k) {
                if ((unsigned)ecol[k] >= (unsigned)n_cols)
                    bad_col = true;
                else
                    ++c[panel_of(ecol[k])];
            }
        }
    }
    if (bad_col) EHYB_FAIL(EHYB_ERR_INTERNAL, "build_panel_residual: column outside the matrix");
    for (int p = 0; p < n_panels; ++p) {
        int64_t at = pcount[p];
        for (int t = 0; t < T; ++t) {
            const int64_t c = run_cnt[(size_t)t * (size_t)n_panels + (size_t)p];
            run_cnt[(size_t)t * (size_t)n_panels + (size_t)p] = at;  // from here on: where run t puts its next entry of panel p
            at += c;
        }
        pcount[p + 1] = at;
    }
    lap("entries counted per run and panel");
    // ---- padded positions: every panel starts on a multiple of 64 entries
    std::vector<int64_t> pstart((size_t)n_panels + 1, 0);
    for (int p = 0; p < n_panels; ++p) pstart[p + 1] = pstart[p] + (pcount[p + 1] - pcount[p] + 63) / 64 * 64;
    const int64_t padded = pstart[n_panels];
    L->pb_val.assign((size_t)padded, 0.0);
    L->pb_col.assign((size_t)padded, 0);
    L->pb_dst.assign((size_t)padded, 0xFFFFFFFFu);
    const bool vmap = !L->er_src.empty();
    L->pb_src.assign(vmap ? (size_t)padded : 0, -1);
    std::vector<int32_t> prow((size_t)padded, -1);  // row of every stored entry, pass-1 order (scratch)
    lap("stream arrays allocated");  // (first touch of fresh pages: most of the time on a freshly started VM)
    // every run deals its entries out: value, local column, row (and the slot-map entry) straight to their final places
    // -- the reads are sequential, the writes go to one open stream per panel
#pragma omp parallel for schedule(static, 1)
    for (int t = 0; t < T; ++t) {
        int64_t* at = &run_cnt[(size_t)t * (size_t)n_panels];
        for (int p = 0; p < n_panels; ++p) at[p] += pstart[p] - pcount[p];  // unpadded -> padded position
        std::vector<uint32_t> idx;
        for (int64_t i = run_first[(size_t)t]; i < run_first[(size_t)t + 1];) {
            // all segments of one row (a long row comes in several, in any order): the row is dealt as a whole
            const int32_t r = (int32_t)(seg_key[(size_t)i] >> 32);
            int64_t i_end = i + 1;
            while (i_end < run_first[(size_t)t + 1] && (int32_t)(seg_key[(size_t)i_end] >> 32) == r) ++i_end;
            auto deal = [&](int64_t k) {
                const int p = panel_of(ecol[k]);
                const int64_t pos = at[p]++;
                L->pb_val[(size_t)pos] = evalv[k];
                L->pb_col[(size_t)pos] = (uint16_t)(ecol[k] - panel_first[p]);
                prow[(size_t)pos] = r;
                if (vmap) L->pb_src[(size_t)pos] = L->er_src[(size_t)k];
            };
            bool ascending = true;
            int32_t last_col = -1;
            for (int64_t q = i; q < i_end && ascending; ++q) {
                const int64_t sg = (int64_t)(seg_key[(size_t)q] & 0xFFFFFFFFull);
                for (int64_t k = L->er_seg_ptr[sg]; k < L->er_seg_ptr[sg + 1] && ascending; ++k) {
                    ascending = last_col <= ecol[k];
                    last_col = ecol[k];
                }
            }
            if (ascending) {
                for (int64_t q = i; q < i_end; ++q) {
                    const int64_t sg = (int64_t)(seg_key[(size_t)q] & 0xFFFFFFFFull);
                    for (int64_t k = L->er_seg_ptr[sg]; k < L->er_seg_ptr[sg + 1]; ++k) deal(k);
                }
            } else {
                idx.clear();
                for (int64_t q = i; q < i_end; ++q) {
                    const int64_t sg = (int64_t)(seg_key[(size_t)q] & 0xFFFFFFFFull);
                    for (int64_t k = L->er_seg_ptr[sg]; k < L->er_seg_ptr[sg + 1]; ++k) idx.push_back((uint32_t)k);
                }
                std::sort(idx.begin(), idx.end(), [&](uint32_t x, uint32_t y) { return ecol[x] != ecol[y] ? ecol[x] < ecol[y] : x < y; });
                for (uint32_t k : idx) deal((int64_t)k);
            }
            i = i_end;
        }
    }
    std::vector<uint64_t>().swap(seg_key);

    lap("entries dealt to the panels");
    // ---- pieces: runs of one row inside a 64-entry chunk of a panel, numbered in pass-1 order
    // piece_of[pos] (temporarily in pb_dst), and per piece its row
    std::vector<int32_t> piece_row;
    {
        // tasks: stretches of one panel's entries, whole 64-entry chunks each (a chunk starts a piece whatever the rows
        // do), so the hub panel of a degree-ordered matrix is shared out like the rest; count first, then fill
        struct Task {
            int64_t b, e;  // padded positions of stored entries
        };
        std::vector<Task> tasks;
        for (int p = 0; p < n_panels; ++p) {
            const int64_t end = pstart[p] + (pcount[p + 1] - pcount[p]);
            for (int64_t b = pstart[p]; b < end; b += 1 << 16) tasks.push_back({b, std::min<int64_t>(b + (1 << 16), end)});
        }
        const int64_t nt = (int64_t)tasks.size();
        std::vector<int64_t> piece0((size_t)nt + 1, 0);
#pragma omp parallel for schedule(dynamic, 4)
        for (int64_t t = 0; t < nt; ++t) {
            int64_t c = 0;
            for (int64_t k = tasks[(size_t)t].b; k < tasks[(size_t)t].e; ++k)
                if ((k & 63) == 0 || prow[(size_t)k] != prow[(size_t)k - 1]) ++c;
            piece0[(size_t)t + 1] = c;
        }
        for (int64_t t = 0; t < nt; ++t) piece0[(size_t)t + 1] += piece0[(size_t)t];
        const int64_t n_pieces = piece0[(size_t)nt];
        if (n_pieces >= 0x7FFFFF00ll) EHYB_FAIL(EHYB_ERR_ARG, "build_panel_residual: too many partial sums");
        piece_row.resize((size_t)n_pieces);
#pragma omp parallel for schedule(dynamic, 4)
        for (int64_t t = 0; t < nt; ++t) {
            int64_t piece = piece0[(size_t)t] - 1;
            for (int64_t k = tasks[(size_t)t].b; k < tasks[(size_t)t].e; ++k) {
                if ((k & 63) == 0 || prow[(size_t)k] != prow[(size_t)k - 1]) piece_row[(size_t)++piece] = prow[(size_t)k];
                L->pb_dst[(size_t)k] = (uint32_t)piece;
            }
        }
    }
    std::vector<int32_t>().swap(prow);
    const int64_t n_pieces = (int64_t)piece_row.size();

    lap("pieces and values");
    // ---- slots: pieces in (row block, panel, pass-1 order) = stable counting sort by row block
    std::vector<int64_t> rb_count((size_t)n_rb + 1, 0);
    for (int64_t j = 0; j < n_pieces; ++j) ++rb_count[(size_t)rb_of_row[piece_row[(size_t)j] - row_begin] + 1];
    for (int b = 0; b < n_rb; ++b) rb_count[b + 1] += rb_count[b];
    std::vector<uint32_t> slot_of_piece((size_t)n_pieces);
    L->pb_row.assign((size_t)n_pieces, 0);
    {
        std::vector<int64_t> fill(rb_count.begin(), rb_count.end() - 1);
        for (int64_t j = 0; j < n_pieces; ++j) {
            const int r = piece_row[(size_t)j];
            const int b = rb_of_row[r - row_begin];
            const int64_t s = fill[b]++;
            slot_of_piece[(size_t)j] = (uint32_t)s;
            L->pb_row[(size_t)s] = (uint16_t)(r - rb_first[b]);
        }
    }
#pragma omp parallel for schedule(static, 65536)
    for (int64_t pos = 0; pos < padded; ++pos)
        if (L->pb_dst[(size_t)pos] != 0xFFFFFFFFu) L->pb_dst[(size_t)pos] = slot_of_piece[L->pb_dst[(size_t)pos]];

    lap("slots");
    const int64_t staged = panel_pass1_items(cfg, G, pstart, nnz_er, L);
    lap("work items");
    // what pass 1 streams: (value, column+flags) + chunk records + jump list instead of a 4-byte slot per entry
    encode_panel_slots(L);
    lap("compressed slots");
    panel_finish(G, rb_count, staged, padded, n_pieces, (int64_t)L->pb_jump.size(), L);
    const int max_rows = L->pb_rows_max;
    if (cfg.verbose)
        printf("panel residual: %lld entries (%lld with padding) in %d panels of %d columns -> %lld partials, %zu + %zu work units, "
               "row blocks <= %d rows\n",
               (long long)nnz_er, (long long)padded, n_panels, W, (long long)n_pieces, L->pb_items1.size() / 2, L->pb_units2.size() / 4,
               max_rows);
    return EHYB_OK;
}

}  // namespace ehyb
