// Host layout builder: permuted row-grouped COO -> EHYB arrays for gfx950.
// Plays the role of COO2EHYB and its helpers (reference convert.c:61-369); new design:
//
//   reference (convert.c)                         here
//   32-row slabs, one CUDA warp each (6,80,107)   64-row slabs, one wave64 each
//   width = max numInRow2 of the slab (111-126)   same rule, counts recomputed here, rounded
//                                                 up to an even number (entries are stored in
//                                                 pairs: one 16-byte value load per lane)
//   window test partStart <= J < partStart+cache  EHYB_WINDOW_REFERENCE: identical test;
//   (247), local id int16 (248)                   EHYB_WINDOW_HALO: own rows, plus the most
//                                                 referenced outside columns gathered into
//                                                 the rest of the LDS window; local id uint16
//   padding (col 0, val 0.0) (269-281)            same
//   residual rows sorted by length, padded to     residual kept as CSR segments sorted by
//   32-row slabs of equal width (148-168)         length (no padding); rows longer than
//                                                 er_seg_len are split (the reference's long-
//                                                 row path, convert.c:33-59,92-101, is broken:
//                                                 SURVEY 8 a-10 item 4)
//   zero residual -> exit(0) (136-139)            supported
//   self-checks exit() (122-125,226-263,287-303)  EHYB_ERR_INTERNAL
//   int sizes                                     64-bit element counts
#include "ehyb_internal.h"

#include <omp.h>

#include <parallel/algorithm>

#include <algorithm>
#include <numeric>

namespace ehyb {

namespace {

struct PartScratch {
    std::vector<int32_t> halo;       // chosen outside columns, ascending
    std::vector<uint32_t> slab_w2;   // pairs per slab
    std::vector<uint32_t> slab_g;    // column-list groups per slab
    std::vector<uint32_t> slab_ner;  // residual pairs per slab (inline form only)
    std::vector<uint8_t> slab_rel;   // 1: the slab stores its columns relative to the lane's own row
};

constexpr int kWideRow = 128;  // ELL entries per row above which the row is multiplied by the residual kernel

inline int halo_lookup(const std::vector<int32_t>& halo, int col)
{
    auto it = std::lower_bound(halo.begin(), halo.end(), col);
    if (it != halo.end() && *it == col) return (int)(it - halo.begin());
    return -1;
}

// Symmetric pair storage for the rows [s, e) of one partition.  For every pair of in-partition
// entries (i,j), (j,i) with bitwise equal values one of the two is kept with the "scatter" mark
// (state 1: the lane of row i also adds a_ij * x_i to row j) and the other is dropped (state 2);
// every other entry stays as it is (state 0), so nothing depends on the matrix being symmetric.
//   * Which side keeps a pair is decided per pair of row GROUPS (consecutive rows with identical
//     column lists -- the d unknowns of a finite-element node): all rows of a group make the same
//     choice, so their stored column lists stay identical and can still be shared.
//   * The choice is an Euler-trail orientation of the group graph: every group keeps half of its
//     pairs, give or take one, so rows that were equally long stay equally long (a slab is as wide
//     as its longest row).
//   * Entries inside a group (the diagonal blocks) stay as they are.
// state is indexed by entry number minus k0.
// partner (may be null): for every kept entry (state 1) the entry it also stands for, indexed like state.
// (Everything lives in flat arrays that a host thread keeps from one partition to the next: one vector per group, and fresh megabyte
// arrays per partition -- mmap, page faults, munmap -- made this function spend more time in the allocator than in its loops.)
struct OrientScratch {
    std::vector<int> grp, gfirst, nb_ptr, nb, seen, inc_ptr, inc_to, inc_id, fill_at, owner, left, at, cptr, crow, cent, cfill, mark, hint, ck, cpos, coff;
    std::vector<uint8_t> mutual;
};
void sym_orient_partition(const matrixCOO* m, const int* rp, int s, int e, int64_t k0, uint8_t* state, int32_t* partner, OrientScratch& W)
{
    const int own = e - s;
    const int* J = m->J;
    const double* V = m->V;
    std::vector<int>&grp = W.grp, &gfirst = W.gfirst;
    grp.resize(own);
    gfirst.clear();
    for (int r = s; r < e; ++r) {
        const int len = rp[r + 1] - rp[r];
        const bool same = r > s && len == rp[r] - rp[r - 1] && (len == 0 || memcmp(J + rp[r], J + rp[r - 1], sizeof(int) * (size_t)len) == 0);
        if (!same) gfirst.push_back(r);
        grp[r - s] = (int)gfirst.size() - 1;
    }
    const int G = (int)gfirst.size();
    // neighbour groups of every group (from its first row), each list sorted and without repeats
    std::vector<int>&nb_ptr = W.nb_ptr, &nb = W.nb, &seen = W.seen;
    nb_ptr.assign(G + 1, 0);
    nb.clear();
    seen.assign(G, -1);
    for (int g = 0; g < G; ++g) {
        const int r = gfirst[g];
        for (int k = rp[r]; k < rp[r + 1]; ++k) {
            if (J[k] < s || J[k] >= e) continue;
            const int h = grp[J[k] - s];
            if (h == g || seen[h] == g) continue;
            seen[h] = g;
            nb.push_back(h);
        }
        std::sort(nb.begin() + nb_ptr[g], nb.end());
        nb_ptr[g + 1] = (int)nb.size();
    }
    auto sees = [&](int h, int g) { return std::binary_search(nb.begin() + nb_ptr[h], nb.begin() + nb_ptr[h + 1], g); };
    // edges = pairs of groups that see each other, numbered in the order (lower group, higher group); incidence lists
    // (other end, edge) sorted by the other end
    std::vector<int>&inc_ptr = W.inc_ptr, &inc_to = W.inc_to, &inc_id = W.inc_id, &fill_at = W.fill_at;
    std::vector<uint8_t>& mutual = W.mutual;  // per neighbour entry (g, h) with h > g: h sees g as well
    inc_ptr.assign(G + 1, 0);
    mutual.assign(nb.size(), 0);
    for (int g = 0; g < G; ++g)
        for (int q = nb_ptr[g]; q < nb_ptr[g + 1]; ++q)
            if (nb[q] > g && sees(nb[q], g)) mutual[q] = 1, ++inc_ptr[g + 1], ++inc_ptr[nb[q] + 1];
    for (int g = 0; g < G; ++g) inc_ptr[g + 1] += inc_ptr[g];
    inc_to.resize(inc_ptr[G]);
    inc_id.resize(inc_ptr[G]);
    fill_at.assign(inc_ptr.begin(), inc_ptr.end() - 1);
    int n_edges = 0;
    for (int g = 0; g < G; ++g)  // (ascending g: the lower neighbours of a group are in place before its own higher ones)
        for (int q = nb_ptr[g]; q < nb_ptr[g + 1]; ++q) {
            const int h = nb[q];
            if (mutual[q]) {
                inc_to[fill_at[g]] = h, inc_id[fill_at[g]++] = n_edges;
                inc_to[fill_at[h]] = g, inc_id[fill_at[h]++] = n_edges;
                ++n_edges;
            }
        }
    std::vector<int>&owner = W.owner, &left = W.left, &at = W.at;
    owner.assign(n_edges, -1);
    left.resize(G);
    at.assign(inc_ptr.begin(), inc_ptr.end() - 1);
    for (int g = 0; g < G; ++g) left[g] = inc_ptr[g + 1] - inc_ptr[g];
    auto walk = [&](int v) {
        for (;;) {
            while (at[v] < inc_ptr[v + 1] && owner[inc_id[at[v]]] >= 0) ++at[v];
            if (at[v] == inc_ptr[v + 1]) return;
            const int u = inc_to[at[v]], id = inc_id[at[v]];
            owner[id] = v;
            --left[v];
            --left[u];
            v = u;
        }
    };
    for (int g = 0; g < G; ++g)
        if (left[g] & 1) walk(g);  // open trails first: they start and end at odd vertices
    for (int g = 0; g < G; ++g)
        while (left[g] > 0) walk(g);  // what is left is Eulerian: closed trails
    // The partner (j, i) of an entry (i, j) is looked for by COLUMN: the in-partition entries of the partition dealt out by column, rows
    // ascending inside a column (entries of one row and column in stored order).  All partners of row i then sit in one short contiguous
    // list -- until round 4 every one of them was searched for in a different row, after a sort of every row's entries by column (the rows of
    // a permuted matrix are not column-sorted): three quarters of this function.
    std::vector<int>&cptr = W.cptr, &crow = W.crow, &cent = W.cent, &cfill = W.cfill;
    cptr.assign(own + 1, 0);
    for (int64_t k = rp[s]; k < rp[e]; ++k)
        if (J[k] >= s && J[k] < e) ++cptr[J[k] - s + 1];
    for (int c = 0; c < own; ++c) cptr[c + 1] += cptr[c];
    crow.resize(cptr[own]);
    cent.resize(cptr[own]);
    cfill.assign(cptr.begin(), cptr.end() - 1);
    for (int r = s; r < e; ++r)
        for (int k = rp[r]; k < rp[r + 1]; ++k)
            if (J[k] >= s && J[k] < e) {
                const int q = cfill[J[k] - s]++;
                crow[q] = r;
                cent[q] = k;
            }
    // (the groups whose pairs the current row's group keeps are MARKED: a look-up instead of a search per entry)
    // and the rows of a group have one column list: entry t of the row above had its partner at hint[t] of that row's column -- with a
    // symmetric pattern the same place in this row's column, checked before any search)
    std::vector<int>&mark = W.mark, &hint = W.hint, &ck = W.ck, &cpos = W.cpos, &coff = W.coff;
    mark.assign(G, -1);
    int marked = -1;
    for (int i = s; i < e; ++i) {
        const int g = grp[i - s];
        if (marked != g) {
            for (int q = inc_ptr[g]; q < inc_ptr[g + 1]; ++q)
                if (owner[inc_id[q]] == g) mark[inc_to[q]] = g;
            marked = g;
        }
        const int* cb = crow.data() + cptr[i - s];
        const int* ce = crow.data() + cptr[i - s + 1];
        const int len = rp[i + 1] - rp[i], clen = (int)(ce - cb);
        if (hint.size() < (size_t)len) hint.resize((size_t)len, 0);
        if (ck.size() < (size_t)len) ck.resize((size_t)len), cpos.resize((size_t)len);
        // first where every candidate's partner would sit -- and a prefetch of its value and state: they are anywhere in the partition's
        // 2-4 MB, one miss after the other was half of this loop --, then the comparisons
        // (which entries of the row are candidates -- in the partition, in a group whose pairs this group keeps -- is the same for every row
        // of a group: found for its first row, kept for the others)
        if (i == gfirst[g]) {
            coff.clear();
            for (int k = rp[i]; k < rp[i + 1]; ++k) {
                const int j = J[k];
                if (j < s || j >= e) continue;
                const int h = grp[j - s];
                if (h != g && mark[h] == g) coff.push_back(k - rp[i]);
            }
        }
        int nc = 0;
        for (const int t : coff) {
            const int k = rp[i] + t, j = J[k];
            if (state[k - k0] != 0) continue;  // (the state of this row's entries changes below only for the entry at hand)
            int o = hint[k - rp[i]];
            if (!(o < clen && cb[o] == j && (o == 0 || cb[o - 1] != j))) o = (int)(std::lower_bound(cb, ce, j) - cb);
            hint[k - rp[i]] = o;
            if (o < clen && cb[o] == j) {
                const int64_t kp = cent[cptr[i - s] + o];
                __builtin_prefetch(V + kp);
                __builtin_prefetch(state + (kp - k0), 1);
                ck[nc] = k;
                cpos[nc++] = o;
            }
        }
        for (int q = 0; q < nc; ++q) {
            const int k = ck[q], j = J[k];
            // the partner (j, i): first unclaimed entry of row j in column i with the same value
            for (int o = cpos[q]; o < clen && cb[o] == j; ++o) {
                const int64_t kp = cent[cptr[i - s] + o];
                if (state[kp - k0] == 0 && V[kp] == V[k]) {
                    state[k - k0] = 1;
                    state[kp - k0] = 2;
                    if (partner) partner[k - k0] = (int32_t)kp;
                    break;
                }
            }
        }
    }
}

}  // namespace

// part_to_er (may be null): partitions (by their index in the layout's own partition list) whose rows go
// to the residual whole -- no window, no halo, zero-width slabs (plan.cpp decides, see ell_pays()).
int build_layout(const matrixCOO* m, int row_begin, int row_end, const Config& cfg, HostLayout* L,
                 const std::vector<uint8_t>* part_to_er, int local_lo, int local_hi, bool defer_panel, bool stats_only)
{
    // multi-GPU (cfg.n_top > 1): the columns a window may hold are the rank's own, [local_lo, local_hi) -- the plan's
    // rows unless the caller lays out a SAMPLE of a rank's partitions (plan.cpp) and names the rank's range
    if (local_lo < 0) local_lo = row_begin, local_hi = row_end;
    if (!m || !L) EHYB_FAIL(EHYB_ERR_ARG, "build_layout: null argument");
    const int n = m->dimension;
    if (n <= 0 || !m->rowIdx || (m->totalNum > 0 && (!m->J || !m->V)))
        EHYB_FAIL(EHYB_ERR_ARG, "build_layout: incomplete matrixCOO (dimension %d)", n);
    if (row_begin < 0 || row_end > n || row_begin >= row_end)
        EHYB_FAIL(EHYB_ERR_ARG, "build_layout: row range [%d,%d) outside [0,%d)", row_begin, row_end, n);
    const int* rp = m->rowIdx;
    if (rp[0] != 0 || rp[n] != m->totalNum)
        EHYB_FAIL(EHYB_ERR_ARG, "build_layout: rowIdx[0]=%d rowIdx[n]=%d totalNum=%d", rp[0], rp[n], m->totalNum);
    for (int i = 0; i < n; ++i)
        if (rp[i + 1] < rp[i]) EHYB_FAIL(EHYB_ERR_ARG, "build_layout: rowIdx not monotone at row %d", i);
    OmpScope omp_scope(cfg.host_threads);

    // Window capacity: two doubles of the LDS budget hold the slab counter of the ELL kernel, so a
    // 10,240-double budget is exactly 80 KiB and two workgroups still fit one CU's 160 KiB.
    const int lds = std::max(kSlabRows, cfg.lds_doubles - 2);
    const bool halo_mode = cfg.window_mode == EHYB_WINDOW_HALO;
    const bool sym = cfg.sym_pairs == 1 && halo_mode;  // symmetric pair storage (in-partition pairs; remote columns are untouched)
    // Direct shape for small matrices (cfg.direct): no window, every row goes to the row-segment kernel,
    // which then assigns y.  bcsstk17's size (11 k rows, 0.4 M entries, 5 MB) is about 170 slabs: a
    // handful of 1024-thread workgroups with 160 KiB windows cannot be spread over 256 CUs, and x is in L2.
    const bool direct = !sym && cfg.n_top <= 1 && row_begin == 0 && row_end == n &&
                        (cfg.direct == 1 || (cfg.direct == 0 && n <= EHYB_DIRECT_MAX_ROWS && cfg.window_mode != EHYB_WINDOW_REFERENCE && cfg.fuse_er != 1));
    L->direct = direct;

    // (cfg.verbose: where the build spends its time)
    double t_lap = wall_seconds();
    auto lap = [&](const char* what) {
        const double now = wall_seconds();
        if (cfg.verbose) printf("layout: %-28s %7.1f ms\n", what, (now - t_lap) * 1e3);
        t_lap = now;
    };
    // ---- partitions: the caller's, cut down to the window capacity where needed
    std::vector<int32_t>& pb = L->part_boundary;
    pb.clear();
    {
        std::vector<int> src;
        // A matrix that never went through the reorder step has nParts from the sizing rule but
        // an all-zero partBoundary: treated as "no partition information".
        // The partitions may end below n: the rows behind them are the empty ghost rows of a
        // rank-local matrix (ehyb_matrix_append_ghosts).
        const bool have_parts = m->partBoundary && m->nParts > 0 && m->partBoundary[m->nParts] > 0 &&
                                m->partBoundary[m->nParts] <= n && m->partBoundary[m->nParts] >= row_end;
        if (have_parts) {
            bool okb = false, oke = false;
            for (int p = 0; p <= m->nParts; ++p) {
                int b = m->partBoundary[p];
                if (b == row_begin) okb = true;
                if (b == row_end) oke = true;
                if (b >= row_begin && b <= row_end) src.push_back(b);
                if (p && b < m->partBoundary[p - 1]) EHYB_FAIL(EHYB_ERR_ARG, "build_layout: partBoundary not monotone");
            }
            if (!okb || !oke)
                EHYB_FAIL(EHYB_ERR_ARG, "build_layout: rows [%d,%d) do not start/end on partition boundaries", row_begin, row_end);
        } else {
            src = {row_begin, row_end};
        }
        const int cap0 = have_parts ? lds / (sym ? 2 : 1) : std::min(lds / (sym ? 2 : 1), cfg.part_rows);
        for (size_t k = 0; k + 1 < src.size(); ++k) {
            int b = src[k], e = src[k + 1];
            if (e == b) continue;  // empty partition
            // The LDS image starts one row below an odd piece start, so such a piece may hold
            // one row less: retry with a smaller cap if a piece would not fit.
            for (int cap = cap0;; --cap) {
                int pieces = (e - b + cap - 1) / cap;
                bool fits = true;
                for (int q = 0; q < pieces && fits; ++q) {
                    int s0 = b + (int)((int64_t)(e - b) * q / pieces), s1 = b + (int)((int64_t)(e - b) * (q + 1) / pieces);
                    fits = ((s0 & 1) + (s1 - s0)) * (sym ? 2 : 1) <= lds;  // symmetric pairs: x image + y accumulators
                }
                if (fits || cap <= 2) {
                    for (int q = 0; q < pieces; ++q) pb.push_back(b + (int)((int64_t)(e - b) * q / pieces));
                    break;
                }
            }
        }
        pb.push_back(row_end);
    }
    const int np = (int)pb.size() - 1;
    L->n_parts = np;
    L->n_cols = n;
    L->row_begin = row_begin;
    L->row_end = row_end;
    const int nrows = row_end - row_begin;

    lap("partitions");
    // ---- symmetric pair storage (cfg.sym_pairs): which entries carry their partner, which are dropped
    const int64_t k0 = rp[row_begin];
    BigVec<uint8_t> state;            // per entry: 0 as it is, 1 kept + scatter, 2 dropped
    std::vector<int32_t> dropped;     // per row
    BigVec<int32_t> partner;          // per kept entry (state 1 only): the dropped entry it also stands for (value map only)
    const bool vmap = cfg.value_map == 1;
    int64_t sym_kept = 0;
    if (sym) {
        // (neither array is written here: every partition zeroes its own stretch of `state` on its own thread, and `partner` is
        // read for kept entries only, which always have one)
        prefault_vector(state, (size_t)(rp[row_end] - k0));
        state.resize((size_t)(rp[row_end] - k0));
        if (vmap) prefault_vector(partner, state.size());
        if (vmap) partner.resize(state.size());
        dropped.assign(nrows, 0);
#pragma omp parallel reduction(+ : sym_kept)
        {
            OrientScratch scratch;
#pragma omp for schedule(dynamic, 2)
            for (int p = 0; p < np; ++p) {
                memset(state.data() + (rp[pb[p]] - k0), 0, (size_t)(rp[pb[p + 1]] - rp[pb[p]]));
                sym_orient_partition(m, rp, pb[p], pb[p + 1], k0, state.data(), vmap ? partner.data() : nullptr, scratch);
                for (int r = pb[p]; r < pb[p + 1]; ++r) {
                    int d = 0;
                    for (int k = rp[r]; k < rp[r + 1]; ++k) d += state[k - k0] == 2, sym_kept += state[k - k0] == 1;
                    dropped[r - row_begin] = d;
                }
            }
        }
    }
    L->sym = sym;
    L->yacc_doubles = 0;

    lap("pair orientation");
    // ---- pass 1: window contents, per-row ELL counts, slab widths
    std::vector<PartScratch> ps(np);
    std::vector<int32_t> cnt_ell(nrows, 0);
    std::vector<uint8_t> lead_row(nrows, 1);
    std::vector<uint8_t> lead_rel(nrows, 1);  // the same for column lists taken RELATIVE to the row (bands, stencils)
    std::vector<uint8_t> row_to_er(nrows, 0);  // whole row in the residual (hub rows)
    // lane order inside a partition: slot t (slab t / 64, lane t % 64) holds row row_at[first + t]
    std::vector<int32_t> row_at(nrows), slot_of(nrows);
    const bool share = cfg.col_sharing != 2;
    L->win_len.assign(np, 0);
    L->part_windowless.assign(np, 0);
    // Called with partitions to give up (plan.cpp, after a first build that ended in the panel form): the panel
    // form is kept for this build too, and the rows of those partitions get their y from its second pass alone
    bool any_windowless = false;
    if (part_to_er && !sym)
        for (uint8_t f : *part_to_er) any_windowless |= f != 0;
    const bool assign_mode = any_windowless && !direct && cfg.er_mode != 1;
    int bad_col = 0, bad_row = 0;
    // cfg.col_map: every host thread keeps ONE array over the columns -- first the count of a partition's outside columns (which of them
    // the window takes), then the mark of the chosen ones -- and puts back what it touched.  Sorting the candidates of every partition
    // and a binary search per outside entry in both passes were a third of the build (audikw_1-like: 60 k candidates, 3,200 chosen, per
    // partition).  The lists below stay for inputs whose column arrays would not fit.
    const bool col_map = halo_mode && col_map_fits(cfg, n);
#pragma omp parallel
    {
        std::vector<int32_t> cand;
        std::vector<std::pair<int32_t, int32_t>> uniq;  // (count, col)
        std::vector<int32_t> dense;                       // by column, hub partitions only (without col_map)
        std::vector<int32_t> cmap;                        // by column, every partition (col_map): all zero between partitions
#pragma omp for schedule(dynamic, 4)
        for (int p = 0; p < np; ++p) {
            const int s = pb[p], e = pb[p + 1];
            const int own = e - s;
            std::vector<int32_t>().swap(dense);
            int wlen;
            PartScratch& S = ps[p];
            const bool whole_to_er = part_to_er && (size_t)p < part_to_er->size() && (*part_to_er)[p] != 0 && !sym;
            if (whole_to_er) {
                wlen = 0;  // nothing of this partition is multiplied from a window: nothing is staged
            } else if (!halo_mode) {
                wlen = std::min(lds - (s & 1), std::min(n, cfg.n_top > 1 ? local_hi : n) - s);
            } else {
                wlen = own;
                // the LDS image starts at the even row below s; with symmetric pairs it is followed
                // by one accumulator per image row
                int hcap = lds - (own + (s & 1)) * (sym ? 2 : 1);
                cand.clear();
                if (col_map && cmap.empty()) cmap.assign((size_t)n, 0);
                for (int r = s; r < e; ++r)
                    for (int k = rp[r]; k < rp[r + 1]; ++k) {
                        int j = m->J[k];
                        if ((unsigned)j >= (unsigned)n) {
                            bad_col = 1;
                            continue;
                        }
                        if (j >= s && j < e) continue;
                        if (cfg.n_top > 1 && (j < local_lo || j >= local_hi)) continue;  // remote column
                        if (!col_map)
                            cand.push_back(j);
                        else if (cmap[(size_t)j]++ == 0)
                            cand.push_back(j);  // (col_map: every candidate once, its count in the array)
                    }
                if (col_map) {
                    if (hcap > 0 && !cand.empty()) {
                        std::sort(cand.begin(), cand.end());
                        uniq.resize(cand.size());
                        for (size_t a = 0; a < cand.size(); ++a) uniq[a] = {cmap[(size_t)cand[a]], cand[a]};
                    } else {
                        uniq.clear();
                    }
                    for (int32_t j : cand) cmap[(size_t)j] = 0;
                }
                if (hcap > 0 && !cand.empty()) {
                    if (!col_map) uniq.clear();
                    if (col_map) {
                        // (counted above)
                    } else if (cand.size() > (size_t)n / 4) {
                        // a hub partition (millions of candidates): counted in a dense array instead of sorted -- the same
                        // (count, column) list, columns ascending (R-MAT 2^24: 25 M candidates, 2 s of sort on one thread)
                        dense.assign((size_t)n, 0);
                        for (int32_t j : cand) ++dense[(size_t)j];
                        for (int j = 0; j < n; ++j)
                            if (dense[(size_t)j]) {
                                uniq.push_back({dense[(size_t)j], j});
                                dense[(size_t)j] = 0;
                            }
                    } else {
                        std::sort(cand.begin(), cand.end());
                        for (size_t a = 0; a < cand.size();) {
                            size_t b = a;
                            while (b < cand.size() && cand[b] == cand[a]) ++b;
                            uniq.push_back({(int32_t)(b - a), cand[a]});
                            a = b;
                        }
                    }
                    if ((int)uniq.size() > hcap) {
                        // most referenced first; ties: lower column
                        std::nth_element(uniq.begin(), uniq.begin() + hcap, uniq.end(),
                                         [](const std::pair<int32_t, int32_t>& x, const std::pair<int32_t, int32_t>& y) {
                                             return x.first != y.first ? x.first > y.first : x.second < y.second;
                                         });
                        uniq.resize(hcap);
                    }
                    S.halo.resize(uniq.size());
                    for (size_t a = 0; a < uniq.size(); ++a) S.halo[a] = uniq[a].second;
                    std::sort(S.halo.begin(), S.halo.end());
                    // (the count below asks 25 M times whether a column was chosen: flags in the same array, where it exists)
                    if (!dense.empty())
                        for (int32_t j : S.halo) dense[(size_t)j] = 1;
                    if (col_map)
                        for (int32_t j : S.halo) cmap[(size_t)j] = 1;
                }
            }
            L->win_len[p] = wlen;
            L->part_windowless[p] = whole_to_er ? 1 : 0;
            const int nslab = (own + kSlabRows - 1) / kSlabRows;
            S.slab_w2.assign(nslab, 0);
            for (int r = s; r < e; ++r) {
                int c = 0;
                for (int k = rp[r]; k < rp[r + 1]; ++k) {
                    int j = m->J[k];
                    if ((unsigned)j >= (unsigned)n) {
                        bad_col = 1;
                        continue;
                    }
                    if (m->I && m->I[k] != r) bad_row = 1;  // convert.c:243-246 "row val check"
                    if (sym && state[k - k0] == 2) continue;  // its partner carries it
                    if (j >= s && j < s + wlen)
                        ++c;
                    else if (halo_mode && !S.halo.empty() &&
                             (col_map ? cmap[(size_t)j] != 0 : dense.empty() ? halo_lookup(S.halo, j) >= 0 : dense[(size_t)j] != 0))
                        ++c;
                }
                // A slab is walked by ONE wave, four pairs per memory round trip: a slab of rows with
                // hundreds of entries keeps a single wave busy for longer than the rest of its
                // workgroup needs for everything else (R-MAT: a 5-slab item of 131 pairs each ended
                // at 116 us of a 120 us launch).  Such rows go to the residual whole, where 64 lanes
                // share a row.
                if ((cfg.hub_rule != 2 && c > kWideRow) || direct || whole_to_er) {
                    row_to_er[r - row_begin] = 1;
                    c = 0;
                }
                cnt_ell[r - row_begin] = c;
            }
            if (col_map)
                for (int32_t j : S.halo) cmap[(size_t)j] = 0;
            // Lane order.  Normally slot t of the partition is row s + t.  With symmetric pairs the
            // lanes add their sums into the LDS accumulators by row index, so the rows of a
            // partition may sit in the slabs in any order: longest stored row first, which makes
            // the rows of a slab equally long (rows with equal column lists are neighbours with
            // equal counts and stay neighbours).
            for (int t = 0; t < own; ++t) row_at[s - row_begin + t] = s + t;
            if (sym)
                std::stable_sort(row_at.begin() + (s - row_begin), row_at.begin() + (e - row_begin),
                                 [&](int a, int b) { return cnt_ell[a - row_begin] > cnt_ell[b - row_begin]; });
            for (int t = 0; t < own; ++t) slot_of[row_at[s - row_begin + t] - row_begin] = t;
            const int32_t* rows_p = &row_at[s - row_begin];  // slot -> row of this partition
            // Slab widths.  A slab is as wide as its longest row, so a few very long rows (R-MAT
            // hubs) would pad 60-odd short rows up to their length.  Per slab the rows are taken
            // longest first and moved to the residual -- whole row, the CSR segments handle any
            // length -- while that lowers the bytes moved: ELL costs 64 x width x ~9 B, a residual
            // entry ~30 B (12 streamed + an uncoalesced 8-byte gather of x).  This is the intent
            // of the reference's long-row path (rows with > 512 in-window entries,
            // convert.c:92-101), which it never launches.
            for (int q = 0; q < nslab; ++q) {
                const int t0 = q * kSlabRows, t1 = std::min(own, t0 + kSlabRows);
                int idx[kSlabRows];
                int m_rows = t1 - t0;
                for (int i = 0; i < m_rows; ++i) idx[i] = rows_p[t0 + i];
                std::sort(idx, idx + m_rows, [&](int a, int b) {
                    int ca = cnt_ell[a - row_begin], cb = cnt_ell[b - row_begin];
                    return ca != cb ? ca > cb : a < b;
                });
                int64_t best_cost = -1, cost0 = 0, moved = 0;
                int best_j = 0;
                for (int j = 0; j <= m_rows && cfg.hub_rule != 2; ++j) {  // rows idx[0..j) go to the residual
                    const int width = j < m_rows ? (cnt_ell[idx[j] - row_begin] + 1) / 2 * 2 : 0;
                    const int64_t cost = moved * 30 + (int64_t)j * 2048 + (int64_t)kSlabRows * width * 9;
                    if (j == 0) cost0 = cost;
                    if (best_cost < 0 || cost < best_cost) best_cost = cost, best_j = j;
                    if (j < m_rows) moved += cnt_ell[idx[j] - row_begin];
                    if (width == 0) break;
                }
                // only a clear win (> 25 % fewer bytes) is worth residual rows: mildly ragged slabs
                // stay pure ELL (an empty residual also saves the second launch)
                if (best_cost * 4 > cost0 * 3) best_j = 0;
                for (int j = 0; j < best_j; ++j) {
                    cnt_ell[idx[j] - row_begin] = 0;
                    row_to_er[idx[j] - row_begin] = 1;
                }
                uint32_t w2 = 0;
                for (int t = t0; t < t1; ++t) w2 = std::max(w2, (uint32_t)(cnt_ell[rows_p[t] - row_begin] + 1) / 2);
                S.slab_w2[q] = w2;
            }
            // Column-list sharing: a row whose column sequence equals that of the row above it
            // (same slab) joins that row's group and stores no column indices of its own.
            // Finite-element matrices with d unknowns per node give groups of d rows.
            // Relative form: a banded or stencil matrix in its natural order has no two rows with the same
            // columns, but row after row with the same OFFSETS from its own diagonal position.  Stored as
            // (column - own row) such rows share their words too; a slab takes whichever form needs fewer
            // groups (bit 7 of its record word 3 says which).  Only for rows whose ELL entries all lie in the
            // partition's own contiguous window (a halo column's place in LDS says nothing about its offset),
            // plain storage.
            S.slab_g.assign(nslab, 0);
            S.slab_rel.assign(nslab, 0);
            std::vector<uint32_t> g_rel(nslab, 0);
            auto own_window_only = [&](int r) {
                for (int k = rp[r]; k < rp[r + 1]; ++k)
                    if (m->J[k] < s || m->J[k] >= s + wlen) return false;
                return true;
            };
            bool prev_inwin = false;
            for (int t = 0; t < own; ++t) {
                const int r = rows_p[t], rprev = t > 0 ? rows_p[t - 1] : r;  // the row in the lane before
                bool lead = true, leadr = true;
                const bool inwin = share && !sym && !row_to_er[r - row_begin] && own_window_only(r);
                if (share && t % kSlabRows != 0 && !row_to_er[r - row_begin] && !row_to_er[rprev - row_begin]) {
                    const int len = rp[r + 1] - rp[r];
                    lead = len != rp[rprev + 1] - rp[rprev] ||
                           (len > 0 && memcmp(m->J + rp[r], m->J + rp[rprev], sizeof(int) * (size_t)len) != 0);
                    // symmetric pairs: the kept / scatter / dropped pattern must be the same as well
                    if (!lead && sym && len > 0 && memcmp(&state[rp[r] - k0], &state[rp[rprev] - k0], (size_t)len) != 0) lead = true;
                    if (inwin && prev_inwin && len == rp[rprev + 1] - rp[rprev]) {
                        leadr = false;
                        for (int k = 0; k < len && !leadr; ++k) leadr = m->J[rp[r] + k] - r != m->J[rp[rprev] + k] - rprev;
                    }
                }
                prev_inwin = inwin;
                lead_row[r - row_begin] = lead ? 1 : 0;
                lead_rel[r - row_begin] = leadr ? 1 : 0;
                S.slab_g[t / kSlabRows] += lead ? 1 : 0;
                g_rel[t / kSlabRows] += leadr ? 1 : 0;
            }
            for (int q = 0; q < nslab; ++q)
                if (g_rel[q] < S.slab_g[q]) {
                    // every row of the slab must qualify for the relative form, not only the sharing ones
                    bool all_in = true;
                    for (int t = q * kSlabRows; t < std::min(own, (q + 1) * kSlabRows) && all_in; ++t)
                        all_in = row_to_er[rows_p[t] - row_begin] || own_window_only(rows_p[t]);
                    if (all_in) {
                        S.slab_rel[q] = 1;
                        S.slab_g[q] = g_rel[q];
                    }
                }
        }
    }
    if (bad_col) EHYB_FAIL(EHYB_ERR_ARG, "build_layout: column index outside [0,%d)", n);
    if (bad_row) EHYB_FAIL(EHYB_ERR_ARG, "build_layout: I[k] does not match the row rowIdx places it in");

    // residual row pointer (row order)
    std::vector<int64_t> er_rp(nrows + 1, 0);
    for (int r = 0; r < nrows; ++r) {
        int len = rp[row_begin + r + 1] - rp[row_begin + r] - (sym ? dropped[r] : 0);
        er_rp[r + 1] = er_rp[r] + (len - cnt_ell[r]);
    }
    const int64_t nnz_er = er_rp[nrows];
    L->part_nnz_ell.assign(np, 0);
    for (int p = 0; p < np; ++p)
        for (int r = pb[p]; r < pb[p + 1]; ++r) L->part_nnz_ell[p] += cnt_ell[r - row_begin];
    const int64_t nnz = (int64_t)rp[row_end] - rp[row_begin];
    const int64_t nnz_ell = nnz - nnz_er;  // entries the ELL part stands for (a kept pair entry counts twice)
    int64_t stored_ell = 0;
    for (int r = 0; r < nrows; ++r) stored_ell += cnt_ell[r];

    lap("pass 1 (windows, widths)");
    // ---- inline form of a tiny residual.  The residual entries of a slab's rows are stored as
    // extra pairs behind the slab's ELL pairs -- values in the same stream, columns as two global
    // 32-bit indices per lane and pair -- and the ELL lanes multiply them straight from global x
    // before writing y: one launch, no read-modify-write of y.  A dependent chain (row pointer ->
    // column -> x) at the end of the slab cost 7 % of the launch when tried; with the slice every
    // address follows from the slab record and only the x gather waits for a load.
    // The CSR segments are built as well, so the two-phase call of the same plan still works.
    L->inline_er = !direct && cfg.n_top <= 1 && nnz_er > 0 && (cfg.fuse_er == 1 || (cfg.fuse_er != 2 && nnz_er * 500 < nnz));
    if (L->inline_er) {
        // every lane of a slab gets as many residual pairs as the slab's longest residual row: fine
        // for a handful of rows (hub rows moved out of the ELL part whole), not for many long ones
        int64_t padded = 0;
        for (int p = 0; p < np && L->inline_er; ++p) {
            PartScratch& S = ps[p];
            S.slab_ner.assign(S.slab_w2.size(), 0);
            for (int r = pb[p]; r < pb[p + 1]; ++r) {
                const int64_t c = er_rp[r - row_begin + 1] - er_rp[r - row_begin];
                uint32_t& ner = S.slab_ner[slot_of[r - row_begin] / kSlabRows];
                ner = std::max<uint32_t>(ner, (uint32_t)std::min<int64_t>((c + 1) / 2, 1 << 20));
                if (ner > 255) L->inline_er = false;  // a long residual row: the CSR kernel is the better tool
            }
            for (uint32_t ner : S.slab_ner) padded += (int64_t)ner * 2 * kSlabRows;
        }
        if (cfg.fuse_er != 1 && padded * 100 > nnz) L->inline_er = false;  // the slices would add > 1 % (x 2: 24 B each) traffic
    }

    // The panel form left to the device (ehyb_plan_create, cfg.symbolic): only where it is certain to be used -- partitions
    // were given up, or the caller asked for it -- because the automatic choice reads the CSR segments this route never
    // builds.  Where EVERY entry is residual (R-MAT: all partitions given up) the device reads the caller's arrays and
    // not even the row-order copies are made.
    L->deferred = HostLayout::Deferred();
    L->pb_host_missing = false;
    const bool defer = defer_panel && cfg.symbolic != 1 && !L->inline_er && !direct && nnz_er > 0 && (cfg.er_mode == 2 || assign_mode);
    const bool view_m = defer && nnz_er == nnz && !sym;

    // ---- pass 2: prefix sums
    L->halo_ptr.assign(np + 1, 0);
    std::vector<int64_t> slab_base(np + 1, 0);
    int max_win = 0;
    for (int p = 0; p < np; ++p) {
        L->halo_ptr[p + 1] = L->halo_ptr[p] + (int32_t)ps[p].halo.size();
        slab_base[p + 1] = slab_base[p] + (int64_t)ps[p].slab_w2.size();
        // the partition's x image (own rows from the even row below the start, then the halo) and,
        // with symmetric pairs, one y accumulator per image row right behind it
        const int image = (pb[p] & 1) + L->win_len[p];
        max_win = std::max(max_win, image + (int)ps[p].halo.size() + (sym ? image : 0));
        if (sym) L->yacc_doubles = std::max(L->yacc_doubles, image);
    }
    if (max_win > lds) EHYB_FAIL(EHYB_ERR_INTERNAL, "build_layout: window of %d doubles exceeds %d", max_win, lds);
    L->lds_doubles = max_win;
    const int64_t nslabs = slab_base[np];
    L->halo_cols.resize(L->halo_ptr[np]);
    L->slab_pair_ptr.assign(nslabs + 1, 0);
    L->slab_row.resize(nslabs);
    L->slab_part.resize(nslabs);
    L->slab_col_ptr.assign(nslabs + 1, 0);
    L->slab_meta.assign((size_t)nslabs * 4, 0);
    int64_t ell_pairs = 0, er_inline_pairs = 0;
    {
        uint64_t acc = 0, acc_c = 0;
        for (int p = 0; p < np; ++p) {
            std::copy(ps[p].halo.begin(), ps[p].halo.end(), L->halo_cols.begin() + L->halo_ptr[p]);
            for (size_t q = 0; q < ps[p].slab_w2.size(); ++q) {
                int64_t sidx = slab_base[p] + (int64_t)q;
                const uint32_t w2 = ps[p].slab_w2[q], g = std::max<uint32_t>(1, ps[p].slab_g[q]);
                const uint32_t ner = L->inline_er ? ps[p].slab_ner[q] : 0;
                L->slab_pair_ptr[sidx] = (uint32_t)acc;
                L->slab_col_ptr[sidx] = (uint32_t)acc_c;
                L->slab_row[sidx] = pb[p] + (int32_t)q * kSlabRows;
                L->slab_part[sidx] = p;
                // what the kernel reads per slab: one 16-byte record
                // (a row has at most one entry per window column, so 2^16 pairs are out of reach
                // unless the input repeats coordinates)
                if (w2 >= (1u << 16)) EHYB_FAIL(EHYB_ERR_ARG, "build_layout: slab wider than 2^17 entries");
                L->slab_meta[4 * sidx + 0] = (uint32_t)acc;
                L->slab_meta[4 * sidx + 1] = (uint32_t)acc_c;
                L->slab_meta[4 * sidx + 2] = (uint32_t)L->slab_row[sidx];
                L->slab_meta[4 * sidx + 3] = (w2 << 16) | (ner << 8) | (ps[p].slab_rel[q] ? 0x80u : 0u) | (g - 1);
                // value stream: w2 ELL pairs then ner residual pairs, 64 lanes x 2 each;
                // column stream: w2 x g shared words then ner x 128 global columns
                acc += w2 + ner;
                acc_c += (uint64_t)w2 * g + (uint64_t)ner * 2 * kSlabRows;
                ell_pairs += w2;
                er_inline_pairs += ner;
                if (acc > 0xFFFFFFFFull || acc_c > 0xFFFFFFFFull)
                    EHYB_FAIL(EHYB_ERR_ARG, "build_layout: ELL part too large for 32-bit offsets");
            }
        }
        L->slab_pair_ptr[nslabs] = (uint32_t)acc;
        L->slab_col_ptr[nslabs] = (uint32_t)acc_c;
    }
    if (stats_only) {
        // (the window sample of plan.cpp: what the windows would hold and cost -- no value is filled, no residual built)
        L->stats.nnz = nnz;
        L->stats.nnz_ell = nnz_ell;
        L->stats.nnz_er = nnz_er;
        L->stats.n_parts = np;
        L->stats.n_slabs = nslabs;
        return EHYB_OK;
    }
    const int64_t size_ell = ell_pairs * 2 * kSlabRows;                   // ELL elements incl. padding
    const int64_t size_stream = (int64_t)L->slab_pair_ptr[nslabs] * 2 * kSlabRows;  // + inline residual pairs
    const int64_t col_words = (int64_t)L->slab_col_ptr[nslabs];

    lap("inline form + prefix sums");
    // ---- pass 3: fill
    // (fresh pages in one sweep: common.cpp; resize() of these does not write -- every partition clears its own stretch below)
    prefault_vector(L->ell_val, (size_t)size_stream), prefault_vector(L->ell_col, (size_t)col_words);
    L->ell_val.resize((size_t)size_stream);
    L->ell_col.resize((size_t)col_words);
    L->lane_group.assign((size_t)nslabs * kSlabRows, 0);
    // symmetric pairs: which row (place in the partition's LDS image) a lane works on; 0xFFFF = none
    L->slab_lrow.assign(sym ? (size_t)nslabs * kSlabRows : 0, (uint16_t)0xFFFF);
    std::vector<int32_t> tcol(view_m ? 0 : (size_t)nnz_er);
    std::vector<double> tval(view_m ? 0 : (size_t)nnz_er);
    // slot maps (cfg.value_map): the entry every slot of a value stream is filled from, so that the numeric
    // phase can be repeated on the device for new values (ehyb_plan_set_values)
    std::vector<int32_t> tsrc(vmap && !view_m ? (size_t)nnz_er : 0);
    if (vmap) prefault_vector(L->ell_src, (size_t)size_stream);
    if (vmap && sym) prefault_vector(L->ell_src2, (size_t)size_stream);
    L->ell_src.resize(vmap ? (size_t)size_stream : 0);
    L->ell_src2.resize(vmap && sym ? (size_t)size_stream : 0);
    L->er_src.clear();
    L->pb_src.clear();
    L->src_entries = m->totalNum;
    int overflow = 0;
#pragma omp parallel
    {
    std::vector<int32_t> hmap;  // cfg.col_map: place + 1 of every outside column of the current partition's window, else 0
#pragma omp for schedule(dynamic, 4)
    for (int p = 0; p < np; ++p) {
        const int s = pb[p], e = pb[p + 1];
        const int wlen = L->win_len[p];
        const PartScratch& S = ps[p];
        if (col_map && !S.halo.empty()) {
            if (hmap.empty()) hmap.assign((size_t)n, 0);
            for (size_t a = 0; a < S.halo.size(); ++a) hmap[(size_t)S.halo[a]] = (int32_t)a + 1;
        }
        // (the column words are OR-ed together: cleared first.  The value stream and its slot maps are written slot by slot -- the entries,
        // then every lane's padding: a fill in advance was half of the bytes this pass wrote)
        std::fill(L->ell_col.begin() + L->slab_col_ptr[slab_base[p]], L->ell_col.begin() + L->slab_col_ptr[slab_base[p + 1]], 0u);
        const bool vmap2 = vmap && sym;
        auto pad_slot = [&](size_t at) {
            L->ell_val[at] = 0.0;
            if (vmap) L->ell_src[at] = -1;
            if (vmap2) L->ell_src2[at] = -1;
        };
        int gid = 0;
        for (int t = 0; t < e - s; ++t) {
            const int r = row_at[s - row_begin + t];
            const int64_t sidx = slab_base[p] + t / kSlabRows;
            const int lane = t % kSlabRows;
            if (sym) L->slab_lrow[(size_t)sidx * kSlabRows + lane] = (uint16_t)(r - (s & ~1));
            const uint64_t pp = L->slab_pair_ptr[sidx];
            const uint32_t w2 = L->slab_meta[4 * sidx + 3] >> 16;
            const uint32_t ner = (L->slab_meta[4 * sidx + 3] >> 8) & 0xFF;
            const uint64_t cp = L->slab_col_ptr[sidx];
            const uint32_t G = (L->slab_meta[4 * sidx + 3] & 0x3F) + 1;
            const bool rel = (L->slab_meta[4 * sidx + 3] & 0x80u) != 0;
            const bool lead = (rel ? lead_rel[r - row_begin] : lead_row[r - row_begin]) != 0;
            gid = lane == 0 ? 0 : gid + (lead ? 1 : 0);
            L->lane_group[(size_t)sidx * kSlabRows + lane] = (uint8_t)gid;
            if (t + 1 == e - s)  // lanes past the last row of the partition read the last group (values 0)
                for (int l2 = lane + 1; l2 < kSlabRows; ++l2) L->lane_group[(size_t)sidx * kSlabRows + l2] = (uint8_t)gid;
            uint32_t k_ell = 0;
            int64_t k_er = view_m ? er_rp[r - row_begin + 1] : er_rp[r - row_begin];  // (view_m: nothing to copy)
            for (int k = view_m ? rp[r + 1] : rp[r]; k < rp[r + 1]; ++k) {
                int j = m->J[k];
                int local = -1;
                const uint8_t st8 = sym ? state[k - k0] : 0;
                if (st8 == 2) continue;  // symmetric pair: stored with its partner
                // window-local index: the LDS image starts at the even row at or below s, so the
                // staging loads of the kernel are 16-byte aligned (x is hipMalloc-aligned)
                if (row_to_er[r - row_begin])
                    local = -1;  // hub row: every entry goes to the residual
                else if (j >= s && j < s + wlen)
                    local = j - (s & ~1);
                else if (halo_mode && !S.halo.empty()) {
                    const int h = col_map ? hmap[(size_t)j] - 1 : halo_lookup(S.halo, j);
                    if (h >= 0) local = (s & 1) + wlen + h;
                }
                if (local >= 0) {
                    if (k_ell >= 2 * w2) {  // convert.c:251-254
                        overflow = 1;
                        continue;
                    }
                    size_t at = (size_t)(((pp + k_ell / 2) * kSlabRows + lane) * 2 + (k_ell & 1));
                    L->ell_val[at] = m->V[k];
                    if (vmap) {
                        L->ell_src[at] = k;
                        if (vmap2) L->ell_src2[at] = st8 == 1 ? partner[k - k0] : -1;
                    }
                    if (st8 == 1) {
                        if (local >= 0x8000 || j < s || j >= e) {  // only own rows have an accumulator
                            overflow = 1;
                            continue;
                        }
                        local |= 0x8000;  // bit 15: also add value * x[row] to row `local`
                    }
                    if (rel) local = (local - (r - (s & ~1))) & 0xFFFF;  // relative to the lane's own place in the LDS image
                    if (lead)  // two 16-bit window-local columns per word, one word per pair and group
                        L->ell_col[(size_t)(cp + (uint64_t)(k_ell / 2) * G + gid)] |= (uint32_t)local << (16 * (k_ell & 1));
                    ++k_ell;
                } else {
                    tcol[(size_t)k_er] = j;
                    tval[(size_t)k_er] = m->V[k];
                    if (vmap) tsrc[(size_t)k_er] = k;
                    if (L->inline_er) {
                        const uint32_t ke = (uint32_t)(k_er - er_rp[r - row_begin]);
                        if (ke >= 2 * ner) {
                            overflow = 1;
                            continue;
                        }
                        const size_t at = (size_t)(((pp + w2 + ke / 2) * kSlabRows + lane) * 2 + (ke & 1));
                        L->ell_val[at] = m->V[k];
                        if (vmap) L->ell_src[at] = k;
                        if (vmap2) L->ell_src2[at] = -1;
                        L->ell_col[(size_t)(cp + (uint64_t)w2 * G + (uint64_t)(ke / 2) * 2 * kSlabRows + (ke & 1) * kSlabRows + lane)] = (uint32_t)j;
                    }
                    ++k_er;
                }
            }
            if (k_er != er_rp[r - row_begin + 1] || (int)k_ell != cnt_ell[r - row_begin]) overflow = 1;
            // the lane's padding: the ELL slots behind its last entry, the inline-residual slots behind its last residual entry
            for (uint32_t q = std::min(k_ell, 2 * w2); q < 2 * w2; ++q) pad_slot((size_t)(((pp + q / 2) * kSlabRows + lane) * 2 + (q & 1)));
            for (uint32_t q = ner ? (uint32_t)std::min<int64_t>(k_er - er_rp[r - row_begin], 2 * ner) : 0; q < 2 * ner; ++q)
                pad_slot((size_t)(((pp + w2 + q / 2) * kSlabRows + lane) * 2 + (q & 1)));
            if (t + 1 == e - s)  // the lanes past the last row of the partition: nothing but padding
                for (int l2 = lane + 1; l2 < kSlabRows; ++l2)
                    for (uint32_t q = 0; q < 2 * (w2 + ner); ++q) pad_slot((size_t)(((pp + q / 2) * kSlabRows + l2) * 2 + (q & 1)));
        }
        if (sym) {
            // Lanes of a group read the same column word, so their mirror products go to the same
            // accumulator: the kernel sums them across lanes first and lets one lane add.  Bits 6-7
            // of the lane's group byte say how: 0/1/2 = add, together with the next 0/1/2 lanes;
            // 3 = another lane adds for this one.  (Runs longer than three are cut into threes.)
            const int64_t s0 = slab_base[p], s1 = slab_base[p + 1];
            for (int64_t sidx = s0; sidx < s1; ++sidx) {
                uint8_t* lg = &L->lane_group[(size_t)sidx * kSlabRows];
                for (int l = 0; l < kSlabRows;) {
                    int run = 1;
                    while (l + run < kSlabRows && lg[l + run] == lg[l]) ++run;
                    for (int q = 0; q < run; ++q) {
                        const int code = q % 3 == 0 ? std::min(2, run - q - 1) : 3;
                        lg[l + q] = (uint8_t)(lg[l + q] | (code << 6));
                    }
                    l += run;
                }
            }
        }
        if (col_map)
            for (int32_t j : S.halo) hmap[(size_t)j] = 0;
    }
    }
    if (overflow) EHYB_FAIL(EHYB_ERR_INTERNAL, "build_layout: entry counts changed between passes");

    lap("pass 3 (fill)");
    // ---- work items: contiguous slab ranges of roughly equal cost = the bytes the ELL launch streams
    // for them.  (An inline residual is part of `pairs`; a residual with a launch of its own costs
    // this one nothing -- charging it here made the ELL phase of R-MAT twice as long: 198 vs 94 us.)
    std::vector<int32_t> item_of_slab(nslabs, 0);
    {
        auto slab_cost = [&](int64_t sidx) {
            const int64_t pairs = L->slab_pair_ptr[sidx + 1] - L->slab_pair_ptr[sidx];
            const int64_t words = L->slab_col_ptr[sidx + 1] - L->slab_col_ptr[sidx];
            if (assign_mode && L->part_windowless[L->slab_part[sidx]]) return (int64_t)0;  // the ELL launch skips it
            return pairs * (kSlabRows * 16) + words * 4 + 1024;
        };
        // Work items: the global slab sequence cut into `want` = items_per_cu x 256 runs of equal cost
        // (2 workgroups are resident per CU, so 512 items are exactly one wave of workgroups; 513
        // would need a second one -- measured 158 us vs 136 us).  A run that crosses a partition
        // boundary becomes several segments, each with its own window.  Cuts within `snap` slabs of
        // a partition boundary move onto it: a window staged for a handful of slabs is wasted.
        // Inside a workgroup the waves take slabs from an LDS counter, so only the byte count of an
        // item matters, not how its slab count divides by the number of waves.
        std::vector<int64_t> prefix(nslabs + 1, 0);
        for (int64_t sidx = 0; sidx < nslabs; ++sidx) prefix[sidx + 1] = prefix[sidx] + slab_cost(sidx);
        const int64_t total = prefix[nslabs];
        const int64_t want = std::max<int64_t>(1, (int64_t)cfg.items_per_cu * kNumCU);
        const int64_t snap = 5;
        std::vector<int64_t> cuts;
        cuts.push_back(0);
        for (int64_t i = 1; i < want && nslabs > 0; ++i) {
            const int64_t goal = total / want * i + total % want * i / want;
            int64_t c = std::lower_bound(prefix.begin(), prefix.end(), goal) - prefix.begin();
            c = std::min<int64_t>(c, nslabs);
            if (c < nslabs) {
                const int p = L->slab_part[c];
                if (c - slab_base[p] < snap)
                    c = slab_base[p];
                else if (slab_base[p + 1] - c < snap)
                    c = slab_base[p + 1];
            }
            if (c > cuts.back() && c < nslabs) cuts.push_back(c);
        }
        if (nslabs > 0) cuts.push_back(nslabs);
        // item i = slabs [cut_lo[i], cut_hi[i])
        std::vector<int64_t> cut_lo(cuts.begin(), cuts.end() - (cuts.empty() ? 0 : 1)), cut_hi(cuts.begin() + (cuts.empty() ? 0 : 1), cuts.end());
        if (sym) {
            // symmetric pairs: the accumulators of a partition's rows live in one workgroup's LDS, so
            // an item is a whole partition (the reorder step makes them equal: nParts = k x 256).
            // Where they are not equal (entry-balanced partitions of a graded mesh, some of them bisected
            // to fit the window) there are more items than CUs: heaviest first, so that the workgroups
            // of the second round are the light ones (workgroups are dispatched in index order).
            cut_lo.clear();
            cut_hi.clear();
            std::vector<int> order;
            for (int p = 0; p < np; ++p)
                if (slab_base[p + 1] > slab_base[p]) order.push_back(p);
            std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
                return prefix[slab_base[a + 1]] - prefix[slab_base[a]] > prefix[slab_base[b + 1]] - prefix[slab_base[b]];
            });
            for (int p : order) {
                cut_lo.push_back(slab_base[p]);
                cut_hi.push_back(slab_base[p + 1]);
            }
        }
        L->items.clear();
        L->segs.clear();
        int64_t window_loads = 0;
        // assign_mode: partitions without a window have no segment at all (the kernel would only walk past them:
        // 1400 such records in one item cost R-MAT 2^24 70 us); a cut range left without a segment joins the item
        // before it (the first ones: the item after them), so the items still tile the slab sequence
        int64_t carry_lo = -1;
        for (size_t i = 0; i < cut_lo.size(); ++i) {
            const int32_t item = (int32_t)(L->items.size() / 8);
            const int32_t seg_begin = (int32_t)(L->segs.size() / 8);
            for (int64_t c = cut_lo[i]; c < cut_hi[i];) {
                const int p = L->slab_part[c];
                const int64_t e = std::min<int64_t>(cut_hi[i], slab_base[p + 1]);
                if (!(assign_mode && L->part_windowless[p])) {
                    const int32_t hb = L->halo_ptr[p], hn = L->halo_ptr[p + 1] - L->halo_ptr[p];
                    const int32_t seg[8] = {p, (int32_t)c, (int32_t)e, hn, pb[p], pb[p + 1], L->win_len[p], hb};
                    L->segs.insert(L->segs.end(), seg, seg + 8);
                    window_loads += L->win_len[p] + hn;
                }
                c = e;
            }
            const bool empty = (int32_t)(L->segs.size() / 8) == seg_begin;
            if (assign_mode && empty) {
                if (!L->items.empty()) {
                    L->items[L->items.size() - 8 + 3] = (int32_t)cut_hi[i];  // the item before takes these slabs
                    for (int64_t t = cut_lo[i]; t < cut_hi[i]; ++t) item_of_slab[t] = item - 1;
                    continue;
                }
                if (i + 1 < cut_lo.size()) {
                    if (carry_lo < 0) carry_lo = cut_lo[i];
                    continue;
                }
                // no partition has a window: one item without segments (the launch does nothing)
            }
            const int64_t lo = carry_lo >= 0 ? carry_lo : cut_lo[i];
            carry_lo = -1;
            for (int64_t t = lo; t < cut_hi[i]; ++t) item_of_slab[t] = item;
            const int32_t rec[8] = {seg_begin, (int32_t)(L->segs.size() / 8), (int32_t)lo, (int32_t)cut_hi[i], 0, 0, 0, 0};
            L->items.insert(L->items.end(), rec, rec + 8);
        }
        L->stats.window_loads = window_loads;
    }
    const int64_t n_items = (int64_t)L->items.size() / 8;

    lap("work items");
    // ---- pass 4: residual segments, grouped by work item, longest first inside an item
    struct Seg {
        int32_t row;
        int32_t item;
        int64_t begin;
        int32_t len;
    };
    std::vector<Seg> segs;
    int64_t rows_er = 0;
    if (defer) {
        for (int rr = 0; rr < nrows; ++rr) rows_er += er_rp[rr + 1] > er_rp[rr];
        HostLayout::Deferred& D = L->deferred;
        D.pending = true;
        D.nnz_er = nnz_er;
        if (view_m) {
            D.col = m->J + k0;
            D.val = m->V + k0;
            D.src = nullptr;
            D.src_base = k0;
            D.want_src = vmap;
        } else {
            D.own_col = std::move(tcol);
            D.own_val = std::move(tval);
            D.own_src = std::move(tsrc);
            D.col = D.own_col.data();
            D.val = D.own_val.data();
            D.src = vmap ? D.own_src.data() : nullptr;
            D.want_src = vmap;
        }
        D.er_rp = std::move(er_rp);
        L->er_seg_ptr.assign(1, 0);
        L->er_seg_row.clear();
        L->er_col.clear();
        L->er_val.clear();
        L->er_blocks.clear();
        L->pb_assign = assign_mode;
        L->er_panel = false;  // until the device has built it
    } else {
        // segments per row first (the partitions tile the rows in order), then every row fills its own: parallel
        auto pieces_of = [&](int64_t len) { return (len == 0 && !direct) ? 0 : (direct ? 1 : (int)((len + cfg.er_seg_len - 1) / cfg.er_seg_len)); };
        std::vector<int64_t> seg_first((size_t)nrows + 1, 0);
#pragma omp parallel for schedule(static, 8192) reduction(+ : rows_er)
        for (int rr = 0; rr < nrows; ++rr) {
            const int64_t len = er_rp[rr + 1] - er_rp[rr];
            // direct shape: the kernel assigns y, so every row has exactly one segment -- an empty one
            // for an empty row (y = 0), an unsplit one for a long row (no atomics on an unzeroed y)
            seg_first[(size_t)rr + 1] = pieces_of(len);
            rows_er += len > 0;
        }
        for (int rr = 0; rr < nrows; ++rr) seg_first[(size_t)rr + 1] += seg_first[(size_t)rr];
        segs.resize((size_t)seg_first[(size_t)nrows]);
#pragma omp parallel for schedule(dynamic, 4)
        for (int p = 0; p < np; ++p)
            for (int r = pb[p]; r < pb[p + 1]; ++r) {
                const int rr = r - row_begin;
                const int64_t len = er_rp[rr + 1] - er_rp[rr];
                const int pieces = pieces_of(len);
                if (pieces == 0) continue;
                const int32_t item = item_of_slab[slab_base[p] + slot_of[rr] / kSlabRows];
                for (int q = 0; q < pieces; ++q) {
                    const int64_t b = er_rp[rr] + len * q / pieces, e2 = er_rp[rr] + len * (q + 1) / pieces;
                    const int32_t row = r | (pieces > 1 ? (int32_t)0x80000000 : 0);
                    segs[(size_t)seg_first[(size_t)rr] + (size_t)q] = {row, item, b, (int32_t)(e2 - b)};
                }
            }
    }
    __gnu_parallel::stable_sort(segs.begin(), segs.end(),
                                [](const Seg& a, const Seg& b) { return a.item != b.item ? a.item < b.item : a.len > b.len; });
    const int64_t nseg = (int64_t)segs.size();
    if (nseg > 0x7FFFFFFFll) EHYB_FAIL(EHYB_ERR_ARG, "build_layout: too many residual segments");
    if (!defer) {
    L->er_seg_ptr.assign(nseg + 1, 0);
    L->er_seg_row.resize(nseg);
    for (int64_t i = 0; i < nseg; ++i) {
        L->er_seg_ptr[i + 1] = L->er_seg_ptr[i] + segs[i].len;
        L->er_seg_row[i] = segs[i].row;
    }
    prefault_vector(L->er_col, (size_t)nnz_er), prefault_vector(L->er_val, (size_t)nnz_er);
    L->er_col.resize((size_t)nnz_er);
    L->er_val.resize((size_t)nnz_er);
    if (vmap) L->er_src.resize((size_t)nnz_er);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < nseg; ++i) {
        if (vmap) std::copy(tsrc.begin() + segs[i].begin, tsrc.begin() + segs[i].begin + segs[i].len, L->er_src.begin() + L->er_seg_ptr[i]);
        std::copy(tcol.begin() + segs[i].begin, tcol.begin() + segs[i].begin + segs[i].len,
                  L->er_col.begin() + L->er_seg_ptr[i]);
        std::copy(tval.begin() + segs[i].begin, tval.begin() + segs[i].begin + segs[i].len,
                  L->er_val.begin() + L->er_seg_ptr[i]);
    }
    {
        // item record words 4..7: segment range and its length bins -- 64 lanes per segment for
        // len >= 128, 16 for 17..127, 4 for <= 16
        int64_t i = 0;
        for (int64_t it = 0; it < n_items; ++it) {
            int32_t* rec = &L->items[(size_t)it * 8];
            rec[4] = (int32_t)i;
            while (i < nseg && segs[i].item == it && segs[i].len >= 128) ++i;
            rec[5] = (int32_t)i;
            while (i < nseg && segs[i].item == it && segs[i].len > 16) ++i;
            rec[6] = (int32_t)i;
            while (i < nseg && segs[i].item == it) ++i;
            rec[7] = (int32_t)i;
        }
        if (i != nseg) EHYB_FAIL(EHYB_ERR_INTERNAL, "build_layout: residual segments not covered by the work items");
    }
    }  // (!defer)
    {
        L->er_bins[0] = 0;
        L->er_bins[3] = (int32_t)nseg;
        // Flat block list for the stand-alone residual kernel (two-launch form): every block of
        // er_threads threads gets one pass worth of same-bin segments {lo, hi, lanes, 0}.
        L->er_blocks.clear();
        const int lanes[3] = {64, 16, 4};
        for (int64_t it = 0; it < n_items; ++it) {
            const int32_t* rec = &L->items[(size_t)it * 8];
            for (int b = 0; b < 3; ++b) {
                const int per = cfg.er_threads / lanes[b];
                for (int32_t lo = rec[4 + b]; lo < rec[5 + b]; lo += per) {
                    const int32_t blk[4] = {lo, std::min(rec[5 + b], lo + per), lanes[b], 0};
                    L->er_blocks.insert(L->er_blocks.end(), blk, blk + 4);
                }
            }
        }
    }
    // the row-order copies are done with: their pages go back before the panel form asks for as many again (on a freshly
    // started VM the first touch of a page costs more than everything the builder does with it afterwards)
    std::vector<int32_t>().swap(tcol);
    std::vector<double>().swap(tval);
    std::vector<int32_t>().swap(tsrc);
    std::vector<Seg>().swap(segs);
    lap("pass 4 (residual segments)");
    // ---- a large residual also gets its panel form (er_panel.cpp): what the residual launch then runs
    ehyb_stats& st = L->stats;
    st.rows_er = rows_er;
    L->er_panel = false;
    // er_mode 0 = automatic: the panel form for a LARGE residual WITHOUT locality.  The CSR kernel
    // gathers x from global memory: where neighbouring rows read neighbouring columns (a structured
    // residual: kkt3d on contiguous partitions) the gathers of a wave fall into a few cache lines and
    // the kernel streams at 5.3 TB/s; where they do not (R-MAT) every entry is its own L2 request and the
    // panel form wins.  Locality is read off the residual itself: distinct 128-byte
    // lines of x per entry over windows of 1024 consecutive residual entries.
    bool want_panel = cfg.er_mode == 2 || assign_mode;
    L->pb_assign = false;
    if (cfg.er_mode == 0 && !assign_mode && !L->inline_er && !direct && nnz_er >= (1 << 21)) {
        const int64_t win = 1024, nwin = std::min<int64_t>(64, nnz_er / win);
        int64_t lines = 0;
        std::vector<int32_t> tmp((size_t)win);
        for (int64_t w = 0; w < nwin; ++w) {
            const int64_t at = (nnz_er - win) * w / std::max<int64_t>(1, nwin - 1);
            for (int64_t k = 0; k < win; ++k) tmp[(size_t)k] = L->er_col[(size_t)(at + k)] >> 4;
            std::sort(tmp.begin(), tmp.end());
            lines += std::unique(tmp.begin(), tmp.end()) - tmp.begin();
        }
        want_panel = lines * 2 > nwin * win;  // more than one new line per two entries: no locality to speak of
        if (cfg.verbose) printf("residual locality: %.3f distinct x lines per entry -> %s form\n", (double)lines / (double)(nwin * win), want_panel ? "panel" : "CSR");
    }
    if (!defer && !L->inline_er && !direct && nnz_er > 0 && want_panel) {
        L->pb_assign = assign_mode;
        const int rc_pb = build_panel_residual(cfg, L);
        if (rc_pb != EHYB_OK) return rc_pb;
        if (!L->er_panel) L->pb_assign = false;
    }
    if (defer) L->pb_assign = assign_mode;
    if (assign_mode && !L->pb_assign) EHYB_FAIL(EHYB_ERR_INTERNAL, "build_layout: partitions were given up but the residual did not end in panel form");
    lap("panel form");
    // ---- statistics (convert.c:140,310; spmv.cu:82)
    st.nnz = nnz;
    st.nnz_ell = nnz_ell;
    st.nnz_er = nnz_er;
    st.size_block_ell = size_ell;
    st.ell_padding = size_ell - stored_ell;
    st.sym_pairs = sym_kept;
    st.size_er = nnz_er;
    st.rows_er = rows_er;
    st.er_segments = nseg;
    st.n_rows = nrows;
    st.n_cols = n;
    st.n_parts = np;
    st.n_slabs = nslabs;
    st.n_items = n_items;
    st.halo_cols = L->halo_ptr[np];
    int maxrow = 0;
    for (int r = row_begin; r < row_end; ++r) maxrow = std::max(maxrow, rp[r + 1] - rp[r]);
    st.max_row = maxrow;
    st.lds_bytes = (int64_t)L->lds_doubles * 8;
    st.bytes_alg = 12 * nnz + 4 * ((int64_t)nrows + 1) + 8 * (int64_t)n + 8 * (int64_t)nrows;
    int64_t halo_item_loads = st.window_loads;
    for (size_t sg = 0; sg < L->segs.size(); sg += 8) halo_item_loads -= L->segs[sg + 6];
    st.col_words = col_words;
    st.er_inline = L->inline_er ? er_inline_pairs * 2 * kSlabRows : 0;
    // values 8 B/element, shared column words 4 B, per slab a 16-byte record + 64-byte lane map;
    // the residual either as inline pairs (their columns are part of col_words) or as CSR segments
    st.bytes_format_ell = 8 * size_ell + 4 * col_words + 80 * nslabs + 32 * st.n_items + 32 * (int64_t)(L->segs.size() / 8) + 8 * st.window_loads +
                          4 * halo_item_loads + 8 * (int64_t)nrows + (L->inline_er ? 8 * st.er_inline + 8 * nnz_er : 0);
    // residual launch: (column, value) streamed, one 8-byte gather of x per entry (at least: a random
    // gather moves a whole sector), per segment its pointer, row and block share, and y read + written
    st.bytes_format = st.bytes_format_ell + (L->inline_er ? 0 : (L->er_panel ? L->pb_bytes : 12 * nnz_er + 8 * nnz_er + 12 * nseg + 16 * nseg));
    st.er_partials = L->er_panel ? L->pb_partials : 0;
    if (nnz_ell + nnz_er != nnz) EHYB_FAIL(EHYB_ERR_INTERNAL, "build_layout: %lld + %lld != %lld", (long long)nnz_ell, (long long)nnz_er, (long long)nnz);
    if (sym) {
        int64_t gone = 0;
        for (int r = 0; r < nrows; ++r) gone += dropped[r];
        // every dropped entry has exactly one kept partner, and the ELL part stands for both
        if (gone != sym_kept || stored_ell + gone != nnz_ell)
            EHYB_FAIL(EHYB_ERR_INTERNAL, "build_layout: symmetric pairs do not add up (%lld kept, %lld dropped)", (long long)sym_kept, (long long)gone);
    }
    if (cfg.verbose) {
        printf("toER is %lld, kernel calculation is %lld\n", (long long)nnz_er, (long long)nnz_ell);
        printf("wasteElement is %lld\n", (long long)st.ell_padding);
        printf("ehyb layout: parts %d slabs %lld items %lld window<=%d doubles, halo cols %lld, residual rows %lld segs %lld\n",
               np, (long long)nslabs, (long long)st.n_items, L->lds_doubles, (long long)st.halo_cols,
               (long long)rows_er, (long long)nseg);
    }
    return EHYB_OK;
}

}  // namespace ehyb
