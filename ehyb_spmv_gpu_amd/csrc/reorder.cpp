// Host pre-step: partition, renumber, permute (the role of reference reordering.c).
//
//   reference                                   here
//   matrixReorder        reordering.c:231-378   ehyb_matrix_reorder(m, 1, cfg)
//   matrixReorder_unsym  reordering.c:41-228    ehyb_matrix_reorder(m, 0, cfg)
//   sortRordrList        reordering.c:18-39     stable per-partition sort below
//
// Same observable contract: reorderList[old] = new, partition-contiguous numbering, rows of
// a partition ordered by their in-partition entry count (descending), I/J/V replaced by the
// permuted row-grouped arrays, numInRow2 = entries inside [partStart, partStart+cache).
// Differences: the partitioner is built in (mt-metis optional, see mtmetis_* below); ties in
// the row sort keep the old order (qsort leaves them unspecified); all scratch is zeroed
// (reordering.c:55,60 reads an un-initialised counter array) and freed.
#include "ehyb_internal.h"

#include <dlfcn.h>
#include <omp.h>

#include <algorithm>
#include <cmath>
#include <numeric>

namespace ehyb {

// ------------------------------------------------------------------ mt-metis (optional)
// Types as built by default in mtmetis.h:52-82 (32-bit ids, float reals).
typedef int (*mtmetis_kway_fn)(const uint32_t* nvtxs, const uint32_t* ncon, const uint32_t* xadj,
                               const uint32_t* adjncy, const int32_t* vwgt, const uint32_t* vsize,
                               const int32_t* adjwgt, const uint32_t* nparts, const float* tpwgts,
                               const float* ubvec, const double* options, int32_t* r_edgecut,
                               uint32_t* where);
typedef double* (*mtmetis_opts_fn)(void);

static mtmetis_kway_fn g_kway = nullptr;
static mtmetis_opts_fn g_opts = nullptr;
static bool g_mtmetis_probed = false;

bool mtmetis_available()
{
    if (!g_mtmetis_probed) {
        g_mtmetis_probed = true;
        void* h = RTLD_DEFAULT;
        if (const char* lib = getenv("EHYB_MTMETIS_LIB")) {
            void* dl = dlopen(lib, RTLD_NOW | RTLD_GLOBAL);
            if (dl) h = dl;
        }
        g_kway = (mtmetis_kway_fn)dlsym(h, "MTMETIS_PartGraphKway");
        g_opts = (mtmetis_opts_fn)dlsym(h, "mtmetis_init_options");
    }
    return g_kway && g_opts;
}

// Same call as reordering.c:270-293: ncon 1, no weights, ubvec 1.001, NTHREADS option.
int mtmetis_partition(int n, const int64_t* xadj, const int* adjncy, int nparts, int nthreads,
                      int* part, int64_t* edgecut)
{
    if (!mtmetis_available()) EHYB_FAIL(EHYB_ERR_STATE, "mt-metis is not linked into this process");
    if (xadj[n] > 0xFFFFFFFFll) EHYB_FAIL(EHYB_ERR_ARG, "mt-metis (32-bit build): too many edges");
    std::vector<uint32_t> x32(n + 1), a32((size_t)xadj[n]);
    for (int i = 0; i <= n; ++i) x32[i] = (uint32_t)xadj[i];
    for (int64_t e = 0; e < xadj[n]; ++e) a32[e] = (uint32_t)adjncy[e];
    uint32_t nv = (uint32_t)n, ncon = 1, np = (uint32_t)nparts;
    float ub = 1.001f;
    double* opts = g_opts();
    opts[2 /* MTMETIS_OPTION_NTHREADS, mtmetis.h:101 */] = nthreads > 0 ? nthreads : 1;
    int32_t cut = 0;
    std::vector<uint32_t> where(n);
    int rc = g_kway(&nv, &ncon, x32.data(), a32.data(), nullptr, nullptr, nullptr, &np, nullptr, &ub,
                    opts, &cut, where.data());
    free(opts);
    if (rc != 1 /* MTMETIS_SUCCESS */) EHYB_FAIL(EHYB_ERR_INTERNAL, "MTMETIS_PartGraphKway returned %d", rc);
    for (int i = 0; i < n; ++i) part[i] = (int)where[i];
    if (edgecut) *edgecut = cut;
    return EHYB_OK;
}

// ------------------------------------------------------------------ adjacency
// Undirected pattern of the matrix without self loops.  symmetric_pattern: every (i,j) has its
// (j,i) stored, so the entries are the adjacency; otherwise both directions are added for
// every entry, as reordering.c:56-89 does (duplicates are harmless: they become edge weight).
static void build_adjacency(const matrixCOO* m, bool symmetric_pattern, std::vector<int64_t>* xadj,
                            std::vector<int>* adj)
{
    const int n = m->dimension;
    const int64_t nnz = m->totalNum;
    xadj->assign((size_t)n + 1, 0);
    if (symmetric_pattern) {
        // every (i,j) has its (j,i) stored and the input is row-grouped (rowIdx delimits the rows): row i of
        // the adjacency is row i of the matrix without its diagonal -- rows are independent
        const int* rp = m->rowIdx;
#pragma omp parallel for schedule(static, 4096)
        for (int i = 0; i < n; ++i) {
            int c = 0;
            for (int k = rp[i]; k < rp[i + 1]; ++k) c += m->J[k] != i;
            (*xadj)[i + 1] = c;
        }
        for (int i = 0; i < n; ++i) (*xadj)[i + 1] += (*xadj)[i];
        adj->resize((size_t)(*xadj)[n]);
#pragma omp parallel for schedule(static, 4096)
        for (int i = 0; i < n; ++i) {
            int64_t at = (*xadj)[i];
            for (int k = rp[i]; k < rp[i + 1]; ++k)
                if (m->J[k] != i) (*adj)[at++] = m->J[k];
        }
        return;
    }
    for (int64_t k = 0; k < nnz; ++k) {
        int i = m->I[k], j = m->J[k];
        if (i == j) continue;
        (*xadj)[i + 1]++;
        (*xadj)[j + 1]++;
    }
    for (int i = 0; i < n; ++i) (*xadj)[i + 1] += (*xadj)[i];
    adj->assign((size_t)(*xadj)[n], 0);
    std::vector<int64_t> fill(xadj->begin(), xadj->end() - 1);
    for (int64_t k = 0; k < nnz; ++k) {
        int i = m->I[k], j = m->J[k];
        if (i == j) continue;
        (*adj)[fill[i]++] = j;
        (*adj)[fill[j]++] = i;
    }
}

// k-way partition of the graph of a symmetric-pattern matrix through its COMPRESSED graph, where that pays: the
// unknowns of one finite-element node have the same column list (the closed neighbourhoods of their vertices are
// equal -- what the layout builder shares column words for, and what METIS calls graph compression), so they are
// one vertex of weight d with 1/d^2 of the edges; a partition never separates them, and the multilevel scheme
// works on a ninth of the edges for 3 unknowns per node (audikw_1).  row_w (may be null): per-row weights to
// balance (entry-balanced partitions); cap in the same unit.  *used = false: no structure to compress (fewer than
// a third of the rows have a twin in the row above), nothing was done.
static int partition_compressed(const matrixCOO* m, const std::vector<int64_t>& xadj, const std::vector<int>& adj, const int* row_w,
                                int nparts, int cap, const Config& c, int* part, int64_t* cut, bool* used, std::vector<int>* group_out)
{
    const int n = m->dimension;
    const int* rp = m->rowIdx;
    *used = false;
    // (no adjacency built -- ehyb_matrix_reorder_blocks leaves it out where this function is tried first: with a symmetric pattern the
    // neighbours of a row are its columns, and the diagonal falls out below as "same group")
    const bool from_rows = adj.empty();
    const int* nbr = from_rows ? m->J : adj.data();
    auto nb_begin = [&](int r) { return from_rows ? (int64_t)rp[r] : xadj[(size_t)r]; };
    auto nb_end = [&](int r) { return from_rows ? (int64_t)rp[r + 1] : xadj[(size_t)r + 1]; };
    std::vector<int> group(n);
    std::vector<int> first;  // first row of every group
    std::vector<uint8_t> same(n, 0);  // the row has the column list of the row above it
#pragma omp parallel for schedule(static, 4096)
    for (int v = 1; v < n; ++v) {
        const int len = rp[v + 1] - rp[v];
        same[v] = len > 0 && len == rp[v] - rp[v - 1] && memcmp(m->J + rp[v], m->J + rp[v - 1], sizeof(int) * (size_t)len) == 0;
    }
    int run = 0;
    for (int v = 0; v < n; ++v) {
        const bool twin = v > 0 && run < 16 && same[v];
        if (!twin) {
            first.push_back(v);
            run = 0;
        }
        ++run;
        group[v] = (int)first.size() - 1;
    }
    const int ng = (int)first.size();
    if ((int64_t)ng * 3 > (int64_t)n * 2 || ng < 2 * nparts) return EHYB_OK;
    std::vector<int64_t> cx((size_t)ng + 1, 0);
    std::vector<int> cw((size_t)ng, 0);
    for (int v = 0; v < n; ++v) cw[group[v]] += row_w ? row_w[v] : 1;
#pragma omp parallel
    {
        std::vector<int> seen((size_t)ng, -1);
#pragma omp for schedule(dynamic, 2048)
        for (int g = 0; g < ng; ++g) {
            const int r = first[g];
            int64_t deg = 0;
            for (int64_t e = nb_begin(r); e < nb_end(r); ++e) {
                const int h = group[nbr[e]];
                if (h != g && seen[h] != g) {
                    seen[h] = g;
                    ++deg;
                }
            }
            cx[(size_t)g + 1] = deg;
        }
    }
    for (int g = 0; g < ng; ++g) cx[(size_t)g + 1] += cx[g];
    std::vector<int> ca((size_t)cx[ng]);
#pragma omp parallel
    {
        std::vector<int> seen((size_t)ng, -1);
#pragma omp for schedule(dynamic, 2048)
        for (int g = 0; g < ng; ++g) {
            const int r = first[g];
            int64_t at = cx[g];
            for (int64_t e = nb_begin(r); e < nb_end(r); ++e) {
                const int h = group[nbr[e]];
                if (h != g && seen[h] != g) {
                    seen[h] = g;
                    ca[(size_t)at++] = h;
                }
            }
        }
    }
    if (c.verbose) printf("compressed graph: %d vertices for %d rows, %lld of %lld edges\n", ng, n, (long long)cx[ng], (long long)(from_rows ? m->totalNum - n : xadj[n]));
    std::vector<int> cpart((size_t)ng, 0);
    const int rc = partition_graph(ng, cx.data(), ca.data(), cw.data(), nparts, cap, c, cpart.data(), cut);
    if (rc != EHYB_OK) return rc;
#pragma omp parallel for schedule(static)
    for (int v = 0; v < n; ++v) part[v] = cpart[group[v]];
    if (c.graph_compress == 3) {
        // the compressed graph as the first coarsening level only: one refinement on the rows themselves, where the unknowns of a node may part
        std::vector<int64_t> rx;
        const int64_t* fx = xadj.data();
        if (from_rows) {
            rx.assign(rp, rp + n + 1);
            fx = rx.data();
        }
        const int rc2 = refine_partition(n, fx, nbr, row_w, nparts, cap, c, part, cut);
        if (rc2 != EHYB_OK) return rc2;
        group.clear();   // (the groups are no longer units of the partition)
    }
    if (group_out) group_out->swap(group);
    *used = true;
    return EHYB_OK;
}

}  // namespace ehyb

using namespace ehyb;

extern "C" int ehyb_matrix_reorder(matrixCOO* m, int symmetric_pattern, const ehyb_config* cfg)
{
    return ehyb_matrix_reorder_blocks(m, symmetric_pattern, cfg, nullptr);
}

extern "C" int ehyb_matrix_reorder_blocks(matrixCOO* m, int symmetric_pattern, const ehyb_config* cfg, int* block_first)
{
    clear_error();
    if (!m || m->dimension <= 0 || m->totalNum < 0 || !m->I || !m->J || !m->V || !m->rowIdx ||
        !m->numInRow || !m->numInRow2 || !m->partBoundary || !m->reorderList)
        EHYB_FAIL(EHYB_ERR_ARG, "ehyb_matrix_reorder: incomplete matrixCOO");
    Config c = resolve_config(cfg);
    OmpScope omp_scope(c.host_threads);
    const int n = m->dimension;
    const int64_t nnz = m->totalNum;
    int nparts = m->nParts;
    if (nparts < 1) EHYB_FAIL(EHYB_ERR_ARG, "ehyb_matrix_reorder: nParts = %d (call ehyb_sizing first)", nparts);
    // The input is row-grouped: rowIdx delimits the rows (solver_test.c:105-124).  Everything below
    // works row by row on that.
    if (m->rowIdx[0] != 0 || m->rowIdx[n] != nnz) EHYB_FAIL(EHYB_ERR_ARG, "ehyb_matrix_reorder: rowIdx does not span the %lld entries", (long long)nnz);
    int64_t bad_entry = -1;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i)
        for (int k = m->rowIdx[i]; k < m->rowIdx[i + 1]; ++k)
            if (m->I[k] != i || (unsigned)m->J[k] >= (unsigned)n) {
#pragma omp critical
                if (bad_entry < 0 || k < bad_entry) bad_entry = k;
            }
    if (bad_entry >= 0)
        EHYB_FAIL(EHYB_ERR_ARG, "ehyb_matrix_reorder: entry %lld (%d,%d) is outside the %d x %d matrix or not in the row rowIdx places it in",
                  (long long)bad_entry, m->I[bad_entry], m->J[bad_entry], n, n);
    if (c.verbose) printf("nParts is %d\n", nparts);

    // boundaries the caller's partBoundary can take (ehyb.h, part_boundary_cap): without a stated
    // capacity the partition count is never raised above the caller's nParts
    const int64_t pb_cap = c.part_boundary_cap > 0 ? c.part_boundary_cap : (int64_t)nparts + 1;
    if (pb_cap < (int64_t)nparts + 1)
        EHYB_FAIL(EHYB_ERR_ARG, "ehyb_matrix_reorder: partBoundary holds %lld ints, nParts = %d needs %d", (long long)pb_cap, nparts, nparts + 1);
    std::vector<int> top_first;  // n_top > 1: first partition of every top-level block

    // cfg.col_map (as in build_layout): one array over the columns per host thread where that fits, instead of sorted lists / hash tables
    const bool col_map = col_map_fits(c, n);
    int cache = m->vectorCacheSize > 0 ? (int)m->vectorCacheSize : c.part_rows;
    int cap = std::max<int64_t>(cache, ((int64_t)n + nparts - 1) / nparts);

    // ---- partition (reordering.c:116-139 / 270-293)
    std::vector<int> part(n, 0);
    std::vector<int> row_order;  // non-empty: the order the rows of a partition are numbered in (degree order)
    {
        std::vector<int64_t> xadj;
        std::vector<int> adj;
        const double ta = wall_seconds();
        // EHYB_PART_DEGREE needs the DEGREES only (entries in the row + entries in the column, the diagonal left out):
        // no adjacency lists -- on R-MAT 2^24 building them was a third of the whole reorder
        const bool degrees_only = c.partitioner == EHYB_PART_DEGREE && c.n_top <= 1;
        // the multilevel scheme on the compressed graph where the rows come in groups with one column list
        // Automatic = with symmetric pair storage only: the partitions of the compressed graph cost the bench matrix with
        // EVERY entry stored 4 % (same-box A/B at equal format bytes, round 2: 1098-1101 against 1123-1166 GFLOP/s;
        // bench.py's plain_storage arm fell from 1094 to 1049) while symmetric pairs gain from them (fewer halo columns).
        const bool compress = (c.graph_compress == 1 || c.graph_compress == 3 || (c.graph_compress == 0 && c.sym_pairs == 1)) && symmetric_pattern != 0 && (c.partitioner == EHYB_PART_AUTO || c.partitioner == EHYB_PART_MULTILEVEL) && n >= 4096;
        // ... and then the compressed graph is made from the rows themselves (symmetric pattern: the columns of a row ARE its neighbours);
        // the adjacency lists -- a copy of J without the diagonal, 311 MB for the bench matrix -- are built only if that attempt declines
        const bool defer_adj = compress && c.n_top <= 1;
        const auto need_adjacency = [&]() {
            if (adj.empty() && xadj.empty()) build_adjacency(m, symmetric_pattern != 0, &xadj, &adj);
        };
        if (defer_adj) {
            // nothing yet
        } else if (degrees_only) {
            xadj.assign((size_t)n + 1, 0);
            std::vector<int> colcnt(n, 0);
            if (!symmetric_pattern) {
#pragma omp parallel for schedule(static, 65536)
                for (int64_t k = 0; k < nnz; ++k)
                    if (m->J[k] != m->I[k]) {
#pragma omp atomic
                        ++colcnt[m->J[k]];
                    }
            }
#pragma omp parallel for schedule(static, 4096)
            for (int i = 0; i < n; ++i) {
                int d = 0;
                for (int k = m->rowIdx[i]; k < m->rowIdx[i + 1]; ++k) d += m->J[k] != i;
                xadj[(size_t)i + 1] = d + colcnt[i];
            }
            for (int i = 0; i < n; ++i) xadj[(size_t)i + 1] += xadj[(size_t)i];
        } else
            build_adjacency(m, symmetric_pattern != 0, &xadj, &adj);
        const double t0 = wall_seconds();
        if (c.verbose) printf("adjacency time is %ld us\n", (long)((t0 - ta) * 1e6));
        int64_t cut = 0;
        int rc;
        if (c.partitioner == EHYB_PART_MTMETIS) {
            rc = mtmetis_partition(n, xadj.data(), adj.data(), nparts, symmetric_pattern ? 1 : 6, part.data(), &cut);
        } else if (c.n_top > 1) {
            // Two levels: n_top row blocks of equal entry counts -- one per GPU, so
            // that every GPU streams the same bytes -- then window-sized partitions inside each
            // block.  The partitions of a block are numbered consecutively.
            std::vector<int> vw(n, 1);
            for (int64_t k = 0; k < nnz; ++k) vw[m->I[k]]++;
            int64_t total = 0;
            int maxw = 1;
            for (int i = 0; i < n; ++i) total += vw[i], maxw = std::max(maxw, vw[i]);
            std::vector<int> top(n, 0);
            int64_t tcap = (int64_t)((double)total / c.n_top * 1.03) + maxw;
            rc = partition_graph(n, xadj.data(), adj.empty() ? nullptr : adj.data(), vw.data(), c.n_top, (int)std::min<int64_t>(tcap, 0x7FFFFFFF), c,
                                 top.data(), &cut);
            if (rc != EHYB_OK) return rc;
            const int usable = std::max(kSlabRows, (int)(cache * 0.97));
            std::vector<int> local(n, -1), verts;
            std::vector<int64_t> sx;
            std::vector<int> sa, spart;
            int offset = 0;
            top_first.assign(c.n_top + 1, 0);
            for (int b = 0; b < c.n_top; ++b) {
                verts.clear();
                for (int i = 0; i < n; ++i)
                    if (top[i] == b) {
                        local[i] = (int)verts.size();
                        verts.push_back(i);
                    }
                const int nb = (int)verts.size();
                const int kb = std::max(1, (nb + usable - 1) / usable);
                sx.assign((size_t)nb + 1, 0);
                sa.clear();
                for (int q = 0; q < nb; ++q) {
                    int v = verts[q];
                    for (int64_t e = xadj[v]; e < xadj[v + 1]; ++e)
                        if (top[adj[e]] == b) sa.push_back(local[adj[e]]);
                    sx[q + 1] = (int64_t)sa.size();
                }
                spart.assign(nb, 0);
                int64_t bcut = 0;
                if (nb > 0) {
                    rc = partition_graph(nb, sx.data(), sa.data(), nullptr, kb, cap, c, spart.data(), &bcut);
                    if (rc != EHYB_OK) return rc;
                }
                for (int q = 0; q < nb; ++q) part[verts[q]] = offset + spart[q];
                top_first[b] = offset;
                offset += kb;
                if ((int64_t)offset + 1 > pb_cap)
                    EHYB_FAIL(EHYB_ERR_ARG, "ehyb_matrix_reorder: the two-level partition needs more than the %lld partBoundary entries "
                                            "the caller holds (set cfg.part_boundary_cap)", (long long)pb_cap);
            }
            top_first[c.n_top] = offset;
            nparts = offset;
            m->nParts = nparts;
        } else {
            // (Symmetric pair storage, one workgroup per partition: partitions balanced on entries
            // instead of rows -- vertex weight = row length, rows free up to cfg.part_rows -- bring the
            // heaviest partition from 8.9 % to 5.3 % above the mean on the audikw_1-like matrix, but the
            // launch gets 1 % slower, 2.7 % on kkt3d-110: the larger partitions pay it back in staging
            // and write-out.  Rows stay the balance criterion.)
            // Exception: symmetric pair storage (one workgroup per partition) on a matrix whose rows differ a
            // lot in length -- a graded mesh: rows of 15 to 300 entries.  Equal ROWS then means partitions of
            // 0.3 to 2.4 times the mean work and the launch waits for the heaviest (1358 instead of ~1800
            // GFLOP/s on the graded audikw_1 stand-in).  There the entries are balanced (vertex weight = row
            // length); partitions of the sparse regions that outgrow the LDS window in rows are bisected
            // by the capacity split below.
            std::vector<int> rowlen;
            std::vector<int> twin_group;  // compressed graph used: the group (node) of every row
            bool weighted = false, by_degree = false;
            if (c.sym_pairs == 1 && n >= 4 * nparts) {
                rowlen.resize(n);
                double sum = 0, sq = 0;
                for (int i = 0; i < n; ++i) {
                    rowlen[i] = std::max(1, m->rowIdx[i + 1] - m->rowIdx[i]);
                    sum += rowlen[i];
                    sq += (double)rowlen[i] * rowlen[i];
                }
                const double mean = sum / n, var = sq / n - mean * mean;
                weighted = var > 0.09 * mean * mean;  // sigma above 30 % of the mean (uniform stand-in: 15 %)
                if (c.balance != 0) weighted = c.balance == 1;  // tuning sweeps
                if (weighted) {
                    int maxw = 1;
                    for (int i = 0; i < n; ++i) maxw = std::max(maxw, rowlen[i]);
                    // Entry-balanced partitions of the SPARSE regions need more rows than a window holds and are
                    // bisected below, so the launch ends up with more items than asked for -- and one item more than
                    // a whole number of rounds of 256 workgroups costs a round (graded audikw_1 stand-in: 257 items).
                    // Ask for fewer: the entry budget W per partition at which the expected count
                    //     sum over rows of max(row length / W, 1 / rows a window holds)
                    // is the number wanted less a margin of 1/64 for the pieces a bisection leaves over.
                    if (nparts >= kNumCU) {
                        const double rows_cap = std::max(64, c.part_rows);
                        const int margin = c.req_margin > 0 ? c.req_margin : (c.req_margin < 0 ? 0 : std::max(2, nparts / 64));
                        const double target = nparts - margin;
                        if ((double)n / rows_cap < target) {
                            double lo = sum / nparts, hi = sum;
                            for (int it = 0; it < 60; ++it) {
                                const double W = 0.5 * (lo + hi);
                                double cnt = 0;
                                for (int i = 0; i < n; ++i) cnt += std::max(rowlen[i] / W, 1.0 / rows_cap);
                                (cnt > target ? lo : hi) = W;
                            }
                            const int want = (int)std::min<double>(nparts, std::max(1.0, std::floor(sum / hi)));
                            if (c.verbose && want != nparts) printf("sparse regions will be bisected: asking for %d partitions instead of %d\n", want, nparts);
                            nparts = want;
                        }
                    }
                    const int64_t wcap = (int64_t)(sum / nparts * 1.03) + maxw;
                    if (c.verbose) printf("row lengths vary (mean %.1f, sigma %.1f): partitions balanced on entries\n", mean, std::sqrt(var));
                    bool done = false;
                    rc = EHYB_OK;
                    if (compress) rc = partition_compressed(m, xadj, adj, rowlen.data(), nparts, (int)std::min<int64_t>(wcap, 0x7FFFFFFF), c, part.data(), &cut, &done, &twin_group);
                    if (rc == EHYB_OK && !done) {
                        if (defer_adj) need_adjacency();
                        rc = partition_graph(n, xadj.data(), adj.empty() ? nullptr : adj.data(), rowlen.data(), nparts, (int)std::min<int64_t>(wcap, 0x7FFFFFFF), c, part.data(), &cut, &by_degree);
                    }
                }
            }
            if (!weighted) {
                bool done = false;
                rc = EHYB_OK;
                if (compress) rc = partition_compressed(m, xadj, adj, nullptr, nparts, cap, c, part.data(), &cut, &done, &twin_group);
                if (rc == EHYB_OK && !done) {
                    if (defer_adj) need_adjacency();
                    rc = partition_graph(n, xadj.data(), adj.empty() ? nullptr : adj.data(), nullptr, nparts, cap, c, part.data(), &cut, &by_degree);
                }
            }
            if (rc == EHYB_OK && by_degree) degree_order(n, xadj.data(), &row_order);
            if (c.verbose) printf("k-way partition time is %ld us\n", (long)((wall_seconds() - t0) * 1e6));
            // Capacity-aware refinement (halo window only): a partition whose own rows plus the
            // distinct outside columns it references do not fit the LDS window would spill
            // entries into the residual.  Such partitions -- and only those -- are bisected, which
            // shrinks both their row count and their halo; on mesh-like matrices the residual
            // (and its kernel launch) disappears.  Partitions that overflow by more than 1.5x
            // (power-law graphs) are left alone: splitting cannot make them fit.
            // (blocks of the degree order: a power-law graph -- bisecting a partition with a graph partitioner finds nothing)
            if (rc == EHYB_OK && c.window_mode == EHYB_WINDOW_HALO && c.cap_split != 2 && !by_degree) {
                const int* rp0 = m->rowIdx;  // the input is row-grouped: row i = entries [rp0[i], rp0[i+1])
                for (int round = 0; round < (weighted ? 8 : 3); ++round) {
                    std::vector<int> demand(nparts, 0), local(n, -1);
                    std::vector<std::vector<int>> members(nparts);
                    for (int i = 0; i < n; ++i) {
                        local[i] = (int)members[part[i]].size();
                        members[part[i]].push_back(i);
                    }
                    // distinct outside columns per partition
#pragma omp parallel
                    {
                        std::vector<int> o;
                        std::vector<int> last;  // cfg.col_map: per column, the last partition of this thread that referenced it
                        if (col_map) last.assign((size_t)n, -1);
#pragma omp for schedule(dynamic, 2)
                        for (int p = 0; p < nparts; ++p) {
                            int distinct = 0;
                            if (col_map) {
                                for (int i : members[p])
                                    for (int k = rp0[i]; k < rp0[i + 1]; ++k) {
                                        const int j = m->J[k];
                                        if (part[j] != p && last[(size_t)j] != p) last[(size_t)j] = p, ++distinct;
                                    }
                            } else {
                                o.clear();
                                for (int i : members[p])
                                    for (int k = rp0[i]; k < rp0[i + 1]; ++k)
                                        if (part[m->J[k]] != p) o.push_back(m->J[k]);
                                std::sort(o.begin(), o.end());
                                distinct = (int)(std::unique(o.begin(), o.end()) - o.begin());
                            }
                            // own rows (twice with symmetric pair storage: x and the y accumulators) + halo
                            demand[p] = (int)members[p].size() * (c.sym_pairs == 1 ? 2 : 1) + 2 + distinct;
                        }
                    }
                    std::vector<int> offenders;
                    for (int p = 0; p < nparts; ++p)
                        if (demand[p] > c.lds_doubles - 2 && (weighted || demand[p] <= c.lds_doubles * 3 / 2) && (int)members[p].size() >= 4 * kSlabRows)
                            offenders.push_back(p);
                    // every split adds a partition: no more of them than the caller's partBoundary can take
                    if ((int64_t)nparts + 1 + (int64_t)offenders.size() > pb_cap)
                        offenders.resize((size_t)std::max<int64_t>(0, pb_cap - nparts - 1));
                    if (offenders.empty()) break;
                    if (c.verbose) printf("capacity split round %d: %zu of %d partitions overflow the window\n", round, offenders.size(), nparts);
                    // bisect the offenders side by side (each on its own induced subgraph), then renumber
                    std::vector<std::vector<int>> halves(offenders.size());
                    int rc_any = EHYB_OK;
                    Config quiet = c;
                    quiet.verbose = 0;
#pragma omp parallel
                    {
                        std::vector<int64_t> sx;
                        std::vector<int> sa;
#pragma omp for schedule(dynamic, 1)
                        for (int oi = 0; oi < (int)offenders.size(); ++oi) {
                            const int p = offenders[oi];
                            const std::vector<int>& verts = members[p];
                            const int nb = (int)verts.size();
                            sx.assign((size_t)nb + 1, 0);
                            sa.clear();
                            for (int q = 0; q < nb; ++q) {
                                int v = verts[q];
                                if (adj.empty()) {  // (adjacency never built: symmetric pattern, the row without its diagonal)
                                    for (int k = rp0[v]; k < rp0[v + 1]; ++k)
                                        if (m->J[k] != v && part[m->J[k]] == p) sa.push_back(local[m->J[k]]);
                                } else {
                                    for (int64_t e = xadj[v]; e < xadj[v + 1]; ++e)
                                        if (part[adj[e]] == p) sa.push_back(local[adj[e]]);
                                }
                                sx[q + 1] = (int64_t)sa.size();
                            }
                            halves[oi].assign(nb, 0);
                            int64_t bcut = 0;
                            int r2 = partition_graph(nb, sx.data(), sa.data(), nullptr, 2, (nb + 1) / 2 + std::max(8, nb / 40), quiet,
                                                     halves[oi].data(), &bcut);
                            if (r2 != EHYB_OK) {
#pragma omp critical
                                rc_any = r2;
                            }
                            // the bisection works on the rows themselves: the unknowns of a node follow its first one
                            if (!twin_group.empty())
                                for (int q = 1; q < nb; ++q)
                                    if (twin_group[verts[q]] == twin_group[verts[q - 1]]) halves[oi][q] = halves[oi][q - 1];
                        }
                    }
                    if (rc_any != EHYB_OK) return rc_any;
                    for (size_t oi = 0; oi < offenders.size(); ++oi) {
                        const std::vector<int>& verts = members[offenders[oi]];
                        for (size_t q = 0; q < verts.size(); ++q)
                            if (halves[oi][q] == 1) part[verts[q]] = nparts;
                        ++nparts;
                    }
                    m->nParts = nparts;
                }
            }
        }
        if (rc != EHYB_OK) return rc;
        if (c.verbose)
            printf("partition time is %ld us, edge cut %lld\n", (long)((wall_seconds() - t0) * 1e6), (long long)cut);
    }

    // (Numbering the partitions for locality -- greedy order over the graph of the partitions, so that the eighth of the
    // work items an XCD takes holds neighbours: 63 % instead of 30 % of the halo columns of the bench matrix then belong to
    // a partition of the same eighth -- was measured with every entry stored: 1056-1066 against 1072-1115 GFLOP/s on the
    // same box.  Not kept: the halo gathers hit the Infinity Cache either way.)
    m->nParts = nparts;  // (may have grown by capacity splits, or shrunk: entry-balanced request for a graded mesh)
    // ---- partition-contiguous numbering in old order (reordering.c:301-321)
    const double t_sort = wall_seconds();
    std::vector<int> part_size(nparts, 0);
    for (int i = 0; i < n; ++i) {
        if ((unsigned)part[i] >= (unsigned)nparts) EHYB_FAIL(EHYB_ERR_INTERNAL, "partitioner returned part %d", part[i]);
        part_size[part[i]]++;
    }
    int* pb = m->partBoundary;
    pb[0] = 0;
    for (int p = 0; p < nparts; ++p) pb[p + 1] = pb[p] + part_size[p];

    // in-partition entry count per old row (reordering.c:327-331)
    std::vector<int> inpart(n, 0);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i) {
        int cnt = 0;
        for (int k = m->rowIdx[i]; k < m->rowIdx[i + 1]; ++k) cnt += part[m->J[k]] == part[i];
        // symmetric pair storage keeps about half of a row's in-partition entries (and all of its
        // halo entries, added below): the sort key is the number of entries the row will STORE
        inpart[i] = c.sym_pairs == 1 ? (cnt + 3) / 2 : cnt;
    }

    if (c.verbose > 1) printf("  row order: in-partition counts %ld us\n", (long)((wall_seconds() - t_sort) * 1e6));
    // rows of each partition, old order, then stable sort by inpart descending (reordering.c:334)
    std::vector<int> rows_of(n);
    {
        std::vector<int> fill(pb, pb + nparts);
        if (row_order.empty())
            for (int i = 0; i < n; ++i) rows_of[fill[part[i]]++] = i;
        else  // blocks of the degree order: the hubs first inside every block too
            for (int r = 0; r < n; ++r) rows_of[fill[part[row_order[r]]]++] = row_order[r];
    }
    if (c.window_mode == EHYB_WINDOW_HALO) {
        // The sort key is "entries the ELL kernel will take".  With a halo window that is the
        // in-partition count plus the entries whose column is among the partition's most
        // referenced outside columns (same selection rule as build_layout).
        // Partitions with millions of candidates first, one at a time with every thread on it: the hub partition of a
        // degree-ordered power-law matrix references a fifth of all entries (R-MAT 2^24: 25 M candidates; one thread counted
        // and looked them up for 1.5 s while the others waited).  Reference counts by column in one dense array (atomic
        // increments), the same selection rule, the chosen columns flagged in the array for the second pass.
        std::vector<uint8_t> done((size_t)nparts, 0);
        {
            std::vector<int> dense;
            std::vector<std::pair<int, int>> uniq;
            const auto by_count = [](const std::pair<int, int>& x, const std::pair<int, int>& y) { return x.first != y.first ? x.first > y.first : x.second < y.second; };
            for (int p = 0; p < nparts; ++p) {
                int64_t entries = 0;
                for (int q = pb[p]; q < pb[p + 1]; ++q) entries += m->rowIdx[rows_of[q] + 1] - m->rowIdx[rows_of[q]];
                if (entries <= ((int64_t)1 << 21)) continue;
                done[(size_t)p] = 1;
                const int own = pb[p + 1] - pb[p];
                const int hcap = c.lds_doubles - 2 - (own + (pb[p] & 1)) * (c.sym_pairs == 1 ? 2 : 1);
                if (hcap <= 0) continue;
                if (dense.empty()) dense.assign((size_t)n, 0);
#pragma omp parallel for schedule(dynamic, 64)
                for (int q = pb[p]; q < pb[p + 1]; ++q) {
                    const int i = rows_of[q];
                    for (int k = m->rowIdx[i]; k < m->rowIdx[i + 1]; ++k)
                        if (m->I[k] == i && part[m->J[k]] != p) {
#pragma omp atomic
                            ++dense[(size_t)m->J[k]];
                        }
                }
                uniq.clear();
                for (int j = 0; j < n; ++j)
                    if (dense[(size_t)j]) {
                        uniq.push_back({dense[(size_t)j], j});
                        dense[(size_t)j] = 0;
                    }
                if ((int)uniq.size() > hcap) {
                    std::nth_element(uniq.begin(), uniq.begin() + hcap, uniq.end(), by_count);
                    uniq.resize((size_t)hcap);
                }
                for (const auto& u : uniq) dense[(size_t)u.second] = -1;  // chosen
#pragma omp parallel for schedule(dynamic, 64)
                for (int q = pb[p]; q < pb[p + 1]; ++q) {
                    const int i = rows_of[q];
                    int add = 0;
                    for (int k = m->rowIdx[i]; k < m->rowIdx[i + 1]; ++k) add += m->I[k] == i && part[m->J[k]] != p && dense[(size_t)m->J[k]] < 0;
                    inpart[i] += add;
                }
                for (const auto& u : uniq) dense[(size_t)u.second] = 0;
            }
        }
        if (c.verbose > 1) printf("  row order: hub partitions done at %ld us\n", (long)((wall_seconds() - t_sort) * 1e6));
#pragma omp parallel
        {
            std::vector<int> cand;
            std::vector<std::pair<int, int>> uniq;
            std::vector<int> hkey, hcnt;
            std::vector<int> cmap;  // cfg.col_map: reference count by column, then -1 = chosen; all zero between partitions
#pragma omp for schedule(dynamic, 1)
            for (int p = 0; p < nparts; ++p) {
                if (done[(size_t)p]) continue;
                const int own = pb[p + 1] - pb[p];
                // 2 doubles hold the kernel's slab counter; symmetric pair storage keeps y accumulators too
                const int hcap = c.lds_doubles - 2 - (own + (pb[p] & 1)) * (c.sym_pairs == 1 ? 2 : 1);
                if (hcap <= 0) continue;
                cand.clear();
                if (col_map) {
                    if (cmap.empty()) cmap.assign((size_t)n, 0);
                    for (int q = pb[p]; q < pb[p + 1]; ++q) {
                        const int i = rows_of[q];
                        for (int k = m->rowIdx[i]; k < m->rowIdx[i + 1]; ++k)
                            if (part[m->J[k]] != p && cmap[(size_t)m->J[k]]++ == 0) cand.push_back(m->J[k]);  // every candidate once (I[k] == i: checked on entry)
                    }
                    if (cand.empty()) continue;
                    uniq.resize(cand.size());
                    for (size_t a = 0; a < cand.size(); ++a) uniq[a] = {cmap[(size_t)cand[a]], cand[a]};
                    for (int j : cand) cmap[(size_t)j] = 0;
                    if ((int)uniq.size() > hcap) {
                        std::nth_element(uniq.begin(), uniq.begin() + hcap, uniq.end(),
                                         [](const std::pair<int, int>& x, const std::pair<int, int>& y) {
                                             return x.first != y.first ? x.first > y.first : x.second < y.second;
                                         });
                        uniq.resize(hcap);
                    }
                    for (const auto& u : uniq) cmap[(size_t)u.second] = -1;  // chosen
                    for (int q = pb[p]; q < pb[p + 1]; ++q) {
                        const int i = rows_of[q];
                        int add = 0;
                        for (int k = m->rowIdx[i]; k < m->rowIdx[i + 1]; ++k) add += cmap[(size_t)m->J[k]] < 0;  // (only outside columns are ever marked)
                        inpart[i] += add;
                    }
                    for (const auto& u : uniq) cmap[(size_t)u.second] = 0;
                    continue;
                }
                for (int q = pb[p]; q < pb[p + 1]; ++q) {
                    int i = rows_of[q];
                    for (int k = m->rowIdx[i]; k < m->rowIdx[i + 1]; ++k)
                        if (m->I[k] == i && part[m->J[k]] != p) cand.push_back(m->J[k]);
                }
                if (cand.empty()) continue;
                uniq.clear();
                // reference counts by column in an open-addressing table of this thread (the candidates of a partition of a
                // matrix without locality are ~10^5 random columns: sorting them was most of this step); the chosen
                // columns keep their slot with the count turned negative, which is what the second pass looks up
                size_t cap = 64;
                while (cap < 2 * cand.size()) cap <<= 1;
                hkey.assign(cap, -1);
                hcnt.assign(cap, 0);
                const size_t mask = cap - 1;
                auto slot_of_col = [&](int col) {
                    size_t h = ((uint32_t)col * 2654435761u) & mask;
                    while (hkey[h] != -1 && hkey[h] != col) h = (h + 1) & mask;
                    return h;
                };
                for (int cc : cand) {
                    const size_t h = slot_of_col(cc);
                    hkey[h] = cc;
                    ++hcnt[h];
                }
                for (size_t h = 0; h < cap; ++h)
                    if (hkey[h] != -1) uniq.push_back({hcnt[h], hkey[h]});
                if ((int)uniq.size() > hcap) {
                    std::nth_element(uniq.begin(), uniq.begin() + hcap, uniq.end(),
                                     [](const std::pair<int, int>& x, const std::pair<int, int>& y) {
                                         return x.first != y.first ? x.first > y.first : x.second < y.second;
                                     });
                    uniq.resize(hcap);
                }
                for (const auto& u : uniq) hcnt[slot_of_col(u.second)] = -1;  // chosen
                for (int q = pb[p]; q < pb[p + 1]; ++q) {
                    int i = rows_of[q];
                    for (int k = m->rowIdx[i]; k < m->rowIdx[i + 1]; ++k)
                        if (m->I[k] == i && part[m->J[k]] != p && hcnt[slot_of_col(m->J[k])] < 0) inpart[i]++;
                }
            }
        }
    }
    if (c.verbose > 1) printf("  row order: halo counts done at %ld us\n", (long)((wall_seconds() - t_sort) * 1e6));
#pragma omp parallel for schedule(dynamic, 8)
    for (int p = 0; p < nparts; ++p)
        std::stable_sort(rows_of.begin() + pb[p], rows_of.begin() + pb[p + 1],
                         [&](int a, int b) { return inpart[a] > inpart[b]; });
    int* list = m->reorderList;
    for (int pos = 0; pos < n; ++pos) list[rows_of[pos]] = pos;

    // ---- permuted CSR (reordering.c:335-362)
    const double t_perm = wall_seconds();
    if (c.verbose) printf("row order time is %ld us\n", (long)((t_perm - t_sort) * 1e6));
    int* num = m->numInRow;
    int* num2 = m->numInRow2;
    int* rp = m->rowIdx;
    // the input is row-grouped (rowIdx delimits the rows, solver_test.c:105-124), so the new row
    // `pos` is the old row rows_of[pos] copied in its stored order: rows are independent
    const std::vector<int> old_rp(rp, rp + n + 1);
    std::fill(num2, num2 + n, 0);
    rp[0] = 0;
    int maxcol = 0;
    for (int pos = 0; pos < n; ++pos) {
        num[pos] = old_rp[rows_of[pos] + 1] - old_rp[rows_of[pos]];
        rp[pos + 1] = rp[pos] + num[pos];
        maxcol = std::max(maxcol, num[pos]);
    }
    // Fresh memory is what this step costs (a first touch of 1.24 GB for the bench matrix takes longer than the gather itself, common.cpp:
    // prefault), so only the values get a new array.  The new COLUMN array is the old row array (the rows are known from rowIdx: nothing
    // reads I here), the new ROW array is the old column array once the columns have been dealt out of it: both are mapped already and
    // exactly as long as needed.  (The arrays of a matrixCOO are malloc()ed by contract -- freed with free(), here and by the
    // reference's driver -- and the caller gets three valid arrays back as before.)
    double* nV = (double*)malloc(sizeof(double) * (size_t)std::max<int64_t>(nnz, 1));
    if (!nV) EHYB_FAIL(EHYB_ERR_ALLOC, "ehyb_matrix_reorder: out of memory for %lld entries", (long long)nnz);
    const double tp0 = wall_seconds();
    prefault(nV, sizeof(double) * (size_t)nnz);
    const double tp1 = wall_seconds();
    int* nJ = m->I;
    const int* oJ = m->J;
#pragma omp parallel for schedule(dynamic, 1024)
    for (int ti = 0; ti < n; ++ti) {
        const int oi = rows_of[ti];
        const int ps = pb[part[oi]];
        int64_t dst = rp[ti];
        int in_window = 0;
        memcpy(nV + dst, m->V + old_rp[oi], sizeof(double) * (size_t)(old_rp[oi + 1] - old_rp[oi]));
        for (int k = old_rp[oi]; k < old_rp[oi + 1]; ++k, ++dst) {
            const int tj = list[oJ[k]];
            nJ[dst] = tj;
            in_window += tj >= ps && tj < ps + cache;
        }
        num2[ti] = in_window;
    }
    int* nI = m->J;
#pragma omp parallel for schedule(static)
    for (int ti = 0; ti < n; ++ti)
        for (int64_t dst = rp[ti]; dst < rp[ti + 1]; ++dst) nI[dst] = ti;
    const double tp2 = wall_seconds();
    release_big(m->V, sizeof(double) * (size_t)nnz);   // (on a helper thread: 74 ms of munmap on the box otherwise, common.cpp)
    m->I = nI;
    m->J = nJ;
    m->V = nV;
    if (c.verbose > 1) printf("  permute: setup %.3f prefault %.3f gather %.3f free %.3f s\n", tp0 - t_perm, tp1 - tp0, tp2 - tp1, wall_seconds() - tp2);
    m->maxCol = maxcol;
    if (c.verbose) printf("permute time is %ld us\n", (long)((wall_seconds() - t_perm) * 1e6));
    if (block_first) {
        if (c.n_top > 1)
            std::copy(top_first.begin(), top_first.end(), block_first);
        else
            block_first[0] = 0, block_first[1] = nparts;
    }
    return EHYB_OK;
}

// Where every entry of the reordered matrix came from: new row `pos` is old row rows_of[pos] copied in its
// stored order (the permuted-CSR step above, reordering.c:348-362), so the entry order is a function of the
// old row pointer and the permutation alone.
extern "C" int ehyb_entry_order(int dimension, const int* row_idx_before, const int* reorder_list, int32_t* entry_order)
{
    clear_error();
    const int n = dimension;
    if (n <= 0 || !row_idx_before || !reorder_list || !entry_order) EHYB_FAIL(EHYB_ERR_ARG, "ehyb_entry_order: bad arguments");
    std::vector<int> old_of(n, -1);
    for (int i = 0; i < n; ++i) {
        const int t = reorder_list[i];
        if ((unsigned)t >= (unsigned)n || old_of[t] >= 0) EHYB_FAIL(EHYB_ERR_ARG, "ehyb_entry_order: reorder_list is not a permutation (entry %d)", i);
        old_of[t] = i;
    }
    std::vector<int64_t> rp_new(n + 1, 0);
    for (int t = 0; t < n; ++t) rp_new[t + 1] = rp_new[t] + (row_idx_before[old_of[t] + 1] - row_idx_before[old_of[t]]);
    if (rp_new[n] != (int64_t)row_idx_before[n] - row_idx_before[0]) EHYB_FAIL(EHYB_ERR_ARG, "ehyb_entry_order: row_idx_before is not a row pointer");
    OmpScope omp(0);
#pragma omp parallel for schedule(static, 1024)
    for (int t = 0; t < n; ++t) {
        const int o = old_of[t];
        int64_t dst = rp_new[t];
        for (int k = row_idx_before[o]; k < row_idx_before[o + 1]; ++k) entry_order[dst++] = k;
    }
    return EHYB_OK;
}

extern "C" int ehyb_top_boundary(const matrixCOO* m, const ehyb_config* cfg, int n_top, int* part_of_block)
{
    if (!m || !part_of_block || n_top < 1) EHYB_FAIL(EHYB_ERR_ARG, "ehyb_top_boundary: bad arguments");
    (void)cfg;
    const int np = m->nParts;
    if (!m->partBoundary || !m->rowIdx || np < 1) EHYB_FAIL(EHYB_ERR_ARG, "ehyb_top_boundary: matrix not reordered");
    // runs of whole partitions with (nearly) equal entry counts
    std::vector<int64_t> w(np + 1, 0);
    for (int p = 0; p < np; ++p)
        w[p + 1] = w[p] + (m->rowIdx[m->partBoundary[p + 1]] - m->rowIdx[m->partBoundary[p]]);
    part_of_block[0] = 0;
    int p = 0;
    for (int b = 1; b < n_top; ++b) {
        int64_t target = w[np] * b / n_top;
        while (p < np && w[p] < target) ++p;
        p = std::max(p, part_of_block[b - 1] + (np >= n_top ? 1 : 0));
        p = std::min(p, np - (np >= n_top ? (n_top - b) : 0));
        part_of_block[b] = p;
    }
    part_of_block[n_top] = np;
    return EHYB_OK;
}
