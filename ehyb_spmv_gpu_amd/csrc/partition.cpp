// Built-in k-way graph partitioner: the stand-in for MTMETIS_PartGraphKway as the reference
// calls it (reordering.c:126-139, 280-293: unit weights, ubvec 1.001, edge-cut objective).
//
// What EHYB needs from the partition is *compact* parts (few columns outside the part's own
// x-segment), each no larger than the LDS window.  Scheme:
//   1. coarsen by heavy-edge matching until ~24 vertices per part remain;
//   2. initial k-way partition of the coarsest graph by "bubble" growing: k seeds spread along
//      a BFS order, all regions grown breadth-first (lightest region first), seeds moved to
//      the region centres and regrown a few times;
//   3. project back level by level with greedy boundary refinement under a weight cap;
//   4. at the finest level enforce the hard cap (rows per part <= window).
// Deterministic for a given seed.  Complexity O(|E|) per level.
#include "ehyb_internal.h"

#include <omp.h>

#include <algorithm>
#include <numeric>
#include <queue>

namespace ehyb {
namespace {

struct GView {
    int n = 0;
    const int64_t* xadj = nullptr;
    const int* adj = nullptr;
    const int* ew = nullptr;  // nullptr: all 1
    const int* vw = nullptr;  // nullptr: all 1
    int wv(int v) const { return vw ? vw[v] : 1; }
    int we(int64_t e) const { return ew ? ew[e] : 1; }
};

struct Graph {
    int n = 0;
    std::vector<int64_t> xadj;
    std::vector<int> adj, ew, vw;
    GView view() const
    {
        GView g;
        g.n = n;
        g.xadj = xadj.data();
        g.adj = adj.data();
        g.ew = ew.data();
        g.vw = vw.data();
        return g;
    }
};

// ------------------------------------------------------------------ coarsening
// Heavy-edge matching followed by contraction.  cmap[v] = coarse vertex of v.
void coarsen_once(const GView& g, int max_vw, uint64_t& rng, Graph* out, std::vector<int>* cmap)
{
    const int n = g.n;
    std::vector<int> match(n, -1);
    // Rounds of mutual proposals first (parallel; the outcome does not depend on the thread count):
    // every unmatched vertex names its heaviest admissible unmatched neighbour, ties broken by a hash
    // of the edge that both ends compute alike, and edges named from both ends are matched.  On a
    // mesh each round settles a third to a half of what is left; the serial greedy pass below then
    // only has the remainder to look at.
    if (n >= 20000) {
        std::vector<int> prop(n, -1);
        const uint64_t salt = splitmix64(rng);
        for (int round = 0; round < 8; ++round) {
#pragma omp parallel for schedule(dynamic, 2048)
            for (int v = 0; v < n; ++v) {
                if (match[v] >= 0) continue;
                int best = -1, bw = -1;
                uint64_t bp = 0;
                const int wvv = g.wv(v);
                for (int64_t e = g.xadj[v]; e < g.xadj[v + 1]; ++e) {
                    const int u = g.adj[e];
                    if (u == v || match[u] >= 0) continue;
                    if (wvv + g.wv(u) > max_vw) continue;
                    const int w = g.we(e);
                    if (w < bw) continue;
                    uint64_t h = ((uint64_t)(uint32_t)std::min(u, v) << 32 | (uint32_t)std::max(u, v)) + salt + (uint64_t)round;
                    const uint64_t p = splitmix64(h);
                    if (w > bw || p > bp) {
                        bw = w;
                        bp = p;
                        best = u;
                    }
                }
                prop[v] = best;
            }
            int64_t matched = 0;
#pragma omp parallel for schedule(static) reduction(+ : matched)
            for (int v = 0; v < n; ++v)
                if (match[v] < 0 && prop[v] >= 0 && prop[prop[v]] == v) {
                    // both ends take this branch; each reads and writes its own match entry only
                    // (prop of an unmatched vertex is from this round)
                    match[v] = prop[v];
                    ++matched;
                }
            if (matched < n / 50) break;
        }
    }
    std::vector<int> order(n);
    std::iota(order.begin(), order.end(), 0);
    for (int i = n - 1; i > 0; --i) {
        int j = (int)(splitmix64(rng) % (uint64_t)(i + 1));
        std::swap(order[i], order[j]);
    }
    for (int idx = 0; idx < n; ++idx) {
        int v = order[idx];
        if (match[v] >= 0) continue;
        int best = -1, bw = -1;
        const int wvv = g.wv(v);
        for (int64_t e = g.xadj[v]; e < g.xadj[v + 1]; ++e) {
            int u = g.adj[e];
            if (u == v || match[u] >= 0) continue;
            if (wvv + g.wv(u) > max_vw) continue;
            int w = g.we(e);
            if (w > bw) {
                bw = w;
                best = u;
            }
        }
        if (best >= 0) {
            match[v] = best;
            match[best] = v;
        } else {
            match[v] = v;
        }
    }
    // (Two-hop matching -- pairing the singles that hang off one hub so that power-law graphs keep
    // coarsening -- was tried: R-MAT scale 22 then goes through 10 levels instead of 2 and the initial
    // partition drops from 9 s to 1 s, but the extra contractions cost 17 s and the residual shrinks
    // by under 1 %.  The matching is left to stall there.)
    cmap->assign(n, -1);
    int nc = 0;
    std::vector<int> first;
    first.reserve(n / 2 + 1);
    for (int v = 0; v < n; ++v) {
        if ((*cmap)[v] >= 0) continue;
        (*cmap)[v] = nc;
        (*cmap)[match[v]] = nc;
        first.push_back(v);
        ++nc;
    }
    // Contraction: the adjacency of coarse vertex c is the merged adjacency of its members.  Coarse
    // vertices are independent: a first parallel pass counts their distinct coarse neighbours, a
    // prefix sum places them, a second pass fills the arrays in place (neighbour order inside a
    // vertex = member order, edge order: the same graph for any thread count; no per-thread
    // buffers to grow and copy).
    out->n = nc;
    out->xadj.assign(nc + 1, 0);
    out->vw.assign(nc, 0);
#pragma omp parallel
    {
        std::vector<int> seen(nc, -1);  // last coarse vertex that counted cu
#pragma omp for schedule(dynamic, 2048)
        for (int c = 0; c < nc; ++c) {
            const int members[2] = {first[c], match[first[c]]};
            const int nm = members[0] == members[1] ? 1 : 2;
            int w = 0;
            int64_t deg = 0;
            for (int k = 0; k < nm; ++k) {
                const int v = members[k];
                w += g.wv(v);
                for (int64_t e = g.xadj[v]; e < g.xadj[v + 1]; ++e) {
                    const int cu = (*cmap)[g.adj[e]];
                    if (cu != c && seen[cu] != c) {
                        seen[cu] = c;
                        ++deg;
                    }
                }
            }
            out->vw[c] = w;
            out->xadj[c + 1] = deg;
        }
    }
    for (int c = 0; c < nc; ++c) out->xadj[c + 1] += out->xadj[c];
    out->adj.resize((size_t)out->xadj[nc]);
    out->ew.resize((size_t)out->xadj[nc]);
#pragma omp parallel
    {
        std::vector<int64_t> pos(nc, -1);  // where coarse neighbour cu sits, if at or behind the vertex's start
#pragma omp for schedule(dynamic, 2048)
        for (int c = 0; c < nc; ++c) {
            const int64_t start = out->xadj[c];
            int64_t at = start;
            const int members[2] = {first[c], match[first[c]]};
            const int nm = members[0] == members[1] ? 1 : 2;
            for (int k = 0; k < nm; ++k) {
                const int v = members[k];
                for (int64_t e = g.xadj[v]; e < g.xadj[v + 1]; ++e) {
                    const int cu = (*cmap)[g.adj[e]];
                    if (cu == c) continue;
                    // entries of earlier vertices sit below `start`, so a stale pos never passes
                    if (pos[cu] >= start && pos[cu] < at && out->adj[pos[cu]] == cu) {
                        out->ew[pos[cu]] += g.we(e);
                    } else {
                        pos[cu] = at;
                        out->adj[at] = cu;
                        out->ew[at] = g.we(e);
                        ++at;
                    }
                }
            }
        }
    }
}

int64_t edge_cut(const GView& g, const int* part)
{
    int64_t cut = 0;
    for (int v = 0; v < g.n; ++v)
        for (int64_t e = g.xadj[v]; e < g.xadj[v + 1]; ++e)
            if (g.adj[e] != v && part[g.adj[e]] != part[v]) cut += g.we(e);
    return cut / 2;
}

// ------------------------------------------------------------------ initial partition
// Breadth-first order of the whole graph (all components), started from a far-away vertex.
void bfs_order(const GView& g, std::vector<int>* order)
{
    const int n = g.n;
    order->clear();
    order->reserve(n);
    std::vector<char> seen(n, 0);
    // two sweeps from vertex 0 to land on a pseudo-peripheral start
    int start = 0;
    for (int sweep = 0; sweep < 2 && n > 0; ++sweep) {
        std::fill(seen.begin(), seen.end(), 0);
        std::vector<int> q;
        q.reserve(n);
        q.push_back(start);
        seen[start] = 1;
        for (size_t h = 0; h < q.size(); ++h) {
            int v = q[h];
            for (int64_t e = g.xadj[v]; e < g.xadj[v + 1]; ++e) {
                int u = g.adj[e];
                if (!seen[u]) {
                    seen[u] = 1;
                    q.push_back(u);
                }
            }
        }
        start = q.back();
    }
    std::fill(seen.begin(), seen.end(), 0);
    int scan = 0;
    for (bool first = true; (int)order->size() < n; first = false) {
        int root = start;
        if (!first) {
            while (seen[scan]) ++scan;  // next unvisited component
            root = scan;
        }
        size_t h = order->size();
        order->push_back(root);
        seen[root] = 1;
        for (; h < order->size(); ++h) {
            int v = (*order)[h];
            for (int64_t e = g.xadj[v]; e < g.xadj[v + 1]; ++e) {
                int u = g.adj[e];
                if (!seen[u]) {
                    seen[u] = 1;
                    order->push_back(u);
                }
            }
        }
    }
}

// Grow all k regions breadth-first from their seeds, always extending the lightest one.
void grow_regions(const GView& g, int k, const std::vector<int>& seeds, int64_t cap,
                  std::vector<int>* part, std::vector<int64_t>* pw)
{
    const int n = g.n;
    part->assign(n, -1);
    pw->assign(k, 0);
    std::vector<std::vector<int>> fr(k);
    std::vector<size_t> head(k, 0);
    using QE = std::pair<int64_t, int>;
    std::priority_queue<QE, std::vector<QE>, std::greater<QE>> heap;
    for (int r = 0; r < k; ++r) {
        if (seeds[r] < 0) continue;
        fr[r].push_back(seeds[r]);
        heap.push({0, r});
    }
    while (!heap.empty()) {
        int r = heap.top().second;
        heap.pop();
        int v = -1;
        while (head[r] < fr[r].size()) {
            int c = fr[r][head[r]++];
            if ((*part)[c] < 0) {
                v = c;
                break;
            }
        }
        if (v < 0) continue;  // region is enclosed
        int w = g.wv(v);
        if ((*pw)[r] + w > cap && (*pw)[r] > 0) continue;  // full: stop growing this region
        (*part)[v] = r;
        (*pw)[r] += w;
        for (int64_t e = g.xadj[v]; e < g.xadj[v + 1]; ++e) {
            int u = g.adj[e];
            if ((*part)[u] < 0) fr[r].push_back(u);
        }
        heap.push({(*pw)[r], r});
    }
    // leftovers (capped-out neighbourhoods, isolated components): lightest adjacent region
    // with room, else the globally lightest region.
    std::vector<int> todo;
    for (int v = 0; v < n; ++v)
        if ((*part)[v] < 0) todo.push_back(v);
    while (!todo.empty()) {
        std::vector<int> next;
        bool progress = false;
        for (int v : todo) {
            int best = -1;
            for (int64_t e = g.xadj[v]; e < g.xadj[v + 1]; ++e) {
                int p = (*part)[g.adj[e]];
                if (p < 0) continue;
                if ((*pw)[p] + g.wv(v) > cap) continue;
                if (best < 0 || (*pw)[p] < (*pw)[best]) best = p;
            }
            if (best >= 0) {
                (*part)[v] = best;
                (*pw)[best] += g.wv(v);
                progress = true;
            } else {
                next.push_back(v);
            }
        }
        if (!progress) {
            for (int v : next) {
                int best = (int)(std::min_element(pw->begin(), pw->end()) - pw->begin());
                (*part)[v] = best;
                (*pw)[best] += g.wv(v);
            }
            next.clear();
        }
        todo.swap(next);
    }
}

// Region centres: the vertex farthest (in hops) from the region's boundary.
void region_centres(const GView& g, int k, const std::vector<int>& part, std::vector<int>* seeds)
{
    const int n = g.n;
    std::vector<int> depth(n, -1);
    std::vector<int> q;
    q.reserve(n);
    for (int v = 0; v < n; ++v) {
        bool boundary = false;
        for (int64_t e = g.xadj[v]; e < g.xadj[v + 1] && !boundary; ++e)
            boundary = part[g.adj[e]] != part[v];
        if (boundary) {
            depth[v] = 0;
            q.push_back(v);
        }
    }
    for (size_t h = 0; h < q.size(); ++h) {
        int v = q[h];
        for (int64_t e = g.xadj[v]; e < g.xadj[v + 1]; ++e) {
            int u = g.adj[e];
            if (depth[u] < 0) {
                depth[u] = depth[v] + 1;
                q.push_back(u);
            }
        }
    }
    std::vector<int> bestd(k, -2);
    seeds->assign(k, -1);
    for (int v = 0; v < n; ++v) {
        int p = part[v];
        int d = depth[v] < 0 ? (1 << 29) : depth[v];  // component without boundary
        if (d > bestd[p]) {
            bestd[p] = d;
            (*seeds)[p] = v;
        }
    }
}

// ------------------------------------------------------------------ refinement
// Greedy boundary refinement: move a vertex to the neighbouring part it is most connected
// to when that lowers the cut (or keeps it and improves balance) and the cap allows it.
int refine_kway(const GView& g, int k, int64_t cap, int passes, std::vector<int>& part,
                std::vector<int64_t>& pw, uint64_t& rng)
{
    const int n = g.n;
    std::vector<int> conn(k, 0);
    std::vector<int> touched;
    touched.reserve(64);
    int total_moves = 0;
    // Only a vertex with a neighbour in another part can move.  The first pass visits the boundary
    // vertices (found in parallel; most vertices are interior and were scanned for nothing before),
    // later passes the vertices whose neighbourhood changed in the pass before.
    std::vector<char> flag(n, 0);
#pragma omp parallel for schedule(static, 4096)
    for (int v = 0; v < n; ++v) {
        const int pv = part[v];
        char b = 0;
        for (int64_t e = g.xadj[v]; e < g.xadj[v + 1] && !b; ++e) b = part[g.adj[e]] != pv;
        flag[v] = b;
    }
    std::vector<int> work, next;
    {
        // cheap shuffle: start at a random offset
        const int off = n ? (int)(splitmix64(rng) % (uint64_t)n) : 0;
        for (int idx = 0; idx < n; ++idx) {
            const int v = idx + off < n ? idx + off : idx + off - n;
            if (flag[v]) work.push_back(v);
        }
    }
    std::fill(flag.begin(), flag.end(), 0);  // from here on: "queued for the next pass"
    // A pass = a parallel FILTER over the work list against a snapshot of the partition (which vertices
    // have a move worth making at all: a positive gain, a tie that improves the balance, or a way out of
    // an overfull part), then the exact greedy step below, serially, for those vertices only.  Most
    // boundary vertices have no such move, so the serial part -- the 1.4 s of the 2.0-2.4 s pre-step of
    // the audikw_1-like matrix in round 1 -- shrinks to the few per cent that do.  A vertex that only
    // becomes movable through a move made earlier in the same pass is queued for the next pass like
    // every neighbour of a moved vertex.  The filter reads only the snapshot: the outcome does not
    // depend on the thread count.
    std::vector<char> cand;
    for (int pass = 0; pass < passes && !work.empty(); ++pass) {
        int moves = 0;
        next.clear();
        const bool filter = work.size() >= 4096;
        if (filter) {
            cand.assign(work.size(), 0);
#pragma omp parallel
            {
                std::vector<int> lconn(k, 0), ltouched;
                ltouched.reserve(64);
#pragma omp for schedule(dynamic, 512)
                for (int64_t wi = 0; wi < (int64_t)work.size(); ++wi) {
                    const int v = work[wi];
                    const int pv = part[v];
                    ltouched.clear();
                    int id = 0;
                    for (int64_t e = g.xadj[v]; e < g.xadj[v + 1]; ++e) {
                        const int u = g.adj[e];
                        if (u == v) continue;
                        const int pu = part[u], w = g.we(e);
                        if (pu == pv)
                            id += w;
                        else {
                            if (lconn[pu] == 0) ltouched.push_back(pu);
                            lconn[pu] += w;
                        }
                    }
                    const int wv = g.wv(v);
                    const bool over = pw[pv] > cap;
                    char c = 0;
                    for (int p : ltouched) {
                        if (!c && pw[p] + wv <= cap + wv) {  // (weights move during the pass: a little slack here, the exact test below)
                            const int gain = lconn[p] - id;
                            c = gain > 0 || (gain == 0 && pw[p] + wv < pw[pv]) || over;
                        }
                        lconn[p] = 0;
                    }
                    cand[wi] = c;
                }
            }
        }
        for (size_t wi = 0; wi < work.size(); ++wi) {
            const int v = work[wi];
            if (filter && !cand[wi]) continue;
            int pv = part[v];
            touched.clear();
            int id = 0;
            for (int64_t e = g.xadj[v]; e < g.xadj[v + 1]; ++e) {
                int u = g.adj[e];
                if (u == v) continue;
                int pu = part[u];
                int w = g.we(e);
                if (pu == pv) {
                    id += w;
                } else {
                    if (conn[pu] == 0) touched.push_back(pu);
                    conn[pu] += w;
                }
            }
            if (touched.empty()) continue;
            int wv = g.wv(v);
            int best = -1;
            for (int p : touched) {
                if (pw[p] + wv > cap) continue;
                if (best < 0 || conn[p] > conn[best] || (conn[p] == conn[best] && pw[p] < pw[best]))
                    best = p;
            }
            if (best >= 0) {
                int gain = conn[best] - id;
                bool over = pw[pv] > cap;
                if (gain > 0 || (gain == 0 && pw[best] + wv < pw[pv]) || (over && pw[best] + wv <= cap)) {
                    part[v] = best;
                    pw[pv] -= wv;
                    pw[best] += wv;
                    ++moves;
                    for (int64_t e = g.xadj[v]; e < g.xadj[v + 1]; ++e) {
                        const int u = g.adj[e];
                        if (!flag[u]) {
                            flag[u] = 1;
                            next.push_back(u);
                        }
                    }
                }
            }
            for (int p : touched) conn[p] = 0;
        }
        total_moves += moves;
        if (moves == 0) break;
        for (int u : next) flag[u] = 0;
        work.swap(next);
    }
    return total_moves;
}

// Hard cap at the finest level: overweight parts shed vertices, least-connected first.
void enforce_cap(const GView& g, int k, int64_t cap, std::vector<int>& part, std::vector<int64_t>& pw)
{
    const int n = g.n;
    bool any = false;
    for (int p = 0; p < k; ++p) any = any || pw[p] > cap;
    if (!any) return;
    std::vector<int> conn(k, 0), touched;
    for (int round = 0; round < 64; ++round) {
        bool over = false;
        for (int v = 0; v < n; ++v) {
            int pv = part[v];
            if (pw[pv] <= cap) continue;
            over = true;
            touched.clear();
            for (int64_t e = g.xadj[v]; e < g.xadj[v + 1]; ++e) {
                int pu = part[g.adj[e]];
                if (pu == pv) continue;
                if (conn[pu] == 0) touched.push_back(pu);
                conn[pu] += g.we(e);
            }
            int wv = g.wv(v);
            int best = -1;
            for (int p : touched) {
                if (pw[p] + wv > cap) continue;
                if (best < 0 || conn[p] > conn[best]) best = p;
            }
            for (int p : touched) conn[p] = 0;
            if (best < 0 && round >= 2) {  // no neighbour has room: any part with room
                for (int p = 0; p < k; ++p)
                    if (pw[p] + wv <= cap && (best < 0 || pw[p] < pw[best])) best = p;
            }
            if (best >= 0) {
                part[v] = best;
                pw[pv] -= wv;
                pw[best] += wv;
            }
        }
        if (!over) break;
    }
}

void initial_partition(const GView& g, int k, int64_t cap, uint64_t& rng, std::vector<int>* part,
                       std::vector<int64_t>* pw)
{
    const int n = g.n;
    std::vector<int> order;
    bfs_order(g, &order);
    std::vector<int> seeds(k, -1);
    // seeds at equal weight intervals along the BFS order
    int64_t total = 0;
    for (int v = 0; v < n; ++v) total += g.wv(v);
    {
        int64_t acc = 0;
        int r = 0;
        for (int i = 0; i < n && r < k; ++i) {
            int64_t mid = (2 * (int64_t)r + 1) * total / (2 * k);
            acc += g.wv(order[i]);
            if (acc > mid) seeds[r++] = order[i];
        }
    }
    std::vector<int> best_part;
    std::vector<int64_t> best_pw;
    int64_t best_cut = -1;
    std::vector<int> cur;
    std::vector<int64_t> cpw;
    // a coarsest graph this large means the matching stalled (power-law graphs): there is little
    // structure for more rounds to find, and each costs seconds
    const int rounds = n > 500000 ? 2 : 6;
    for (int it = 0; it < rounds; ++it) {
        grow_regions(g, k, seeds, cap, &cur, &cpw);
        refine_kway(g, k, cap, 2, cur, cpw, rng);
        int64_t cut = edge_cut(g, cur.data());
        if (best_cut < 0 || cut < best_cut) {
            best_cut = cut;
            best_part = cur;
            best_pw = cpw;
        }
        if (it + 1 < rounds) region_centres(g, k, cur, &seeds);
    }
    part->swap(best_part);
    pw->swap(best_pw);
}

}  // namespace

// Vertices in order of falling degree, ties in the given numbering (EHYB_PART_DEGREE).
void degree_order(int n, const int64_t* xadj, std::vector<int>* order)
{
    // falling degree, ties in the given numbering: a counting sort (R-MAT 2^24: std::stable_sort of 16.8 M rows took 1.5 s)
    order->resize((size_t)n);
    int64_t maxd = 0;
    for (int v = 0; v < n; ++v) maxd = std::max(maxd, xadj[v + 1] - xadj[v]);
    std::vector<int64_t> first((size_t)maxd + 2, 0);  // first[d] = rows of degree > d, after the prefix sum
    for (int v = 0; v < n; ++v) ++first[(size_t)(xadj[v + 1] - xadj[v])];
    int64_t before = 0;
    for (int64_t d = maxd; d >= 0; --d) {
        const int64_t c = first[(size_t)d];
        first[(size_t)d] = before;
        before += c;
    }
    for (int v = 0; v < n; ++v) (*order)[(size_t)first[(size_t)(xadj[v + 1] - xadj[v])]++] = v;
}

int partition_graph(int n, const int64_t* xadj, const int* adjncy, const int* vwgt, int nparts,
                    int max_part_w, const Config& cfg, int* part, int64_t* edgecut, bool* by_degree)
{
    if (by_degree) *by_degree = false;
    // (EHYB_PART_DEGREE reads the degrees only: adjncy may be null there, and the edge cut is then not computed)
    if (n < 0 || nparts < 1 || !xadj || (!adjncy && xadj[n] > 0 && cfg.partitioner != EHYB_PART_DEGREE) || !part)
        EHYB_FAIL(EHYB_ERR_ARG, "partition_graph: bad arguments (n=%d, nparts=%d)", n, nparts);
    if (edgecut) *edgecut = 0;
    if (n == 0) return EHYB_OK;
    int64_t total = 0;
    int max_v = 1;
    for (int v = 0; v < n; ++v) {
        int w = vwgt ? vwgt[v] : 1;
        total += w;
        max_v = std::max(max_v, w);
    }
    int64_t cap = max_part_w > 0 ? max_part_w : (int64_t)((double)total / nparts * 1.001) + 1;
    if (cap * nparts < total)
        EHYB_FAIL(EHYB_ERR_ARG, "partition_graph: %d parts of at most %lld cannot hold weight %lld",
                  nparts, (long long)cap, (long long)total);
    if (nparts == 1) {
        std::fill(part, part + n, 0);
        return EHYB_OK;
    }

    uint64_t rng = 0x1234ABCDull + (uint64_t)cfg.seed * 0x9E3779B97F4A7C15ull;
    GView fine;
    fine.n = n;
    fine.xadj = xadj;
    fine.adj = adjncy;
    fine.ew = nullptr;
    fine.vw = vwgt;

    // contiguous blocks of the given numbering.  Unit weights: equal chunks rounded up to whole
    // 64-row slabs (so block-structured inputs keep their alignment: config 3's 1024-row blocks
    // stay inside one window); weighted: equal-weight blocks.
    // ord (may be null): the same blocks along another order of the vertices (degree order)
    auto contiguous = [&](const int* ord = nullptr) -> int {
        auto at = [&](int r) { return ord ? ord[r] : r; };
        if (!vwgt) {
            // fill every block to the cap (rounded down to whole slabs): block-structured inputs whose
            // block size divides the cap keep whole blocks inside one window
            int64_t chunk = cap >= kSlabRows ? cap / kSlabRows * kSlabRows : cap;
            if (chunk * nparts < n) chunk = cap;
            for (int v = 0; v < n; ++v) part[at(v)] = (int)std::min<int64_t>(v / chunk, nparts - 1);
            int64_t last = n - chunk * (nparts - 1);
            if (last > cap) {  // rounding down to the cap left too much for the last block: spread evenly
                for (int v = 0; v < n; ++v) part[at(v)] = (int)((int64_t)v * nparts / n);
            }
            if (edgecut) *edgecut = adjncy ? edge_cut(fine, part) : -1;
            return EHYB_OK;
        }
        int64_t acc = 0;
        int p = 0;
        int64_t in_p = 0;
        for (int r = 0; r < n; ++r) {
            const int v = at(r);
            int w = fine.wv(v);
            int64_t target = (total * (p + 1) + nparts - 1) / nparts;
            if ((acc + w > target || in_p + w > cap) && p + 1 < nparts && in_p > 0) {
                ++p;
                in_p = 0;
            }
            part[v] = p;
            acc += w;
            in_p += w;
        }
        if (edgecut) *edgecut = adjncy ? edge_cut(fine, part) : -1;
        return EHYB_OK;
    };
    if (cfg.partitioner == EHYB_PART_CONTIGUOUS) return contiguous();
    // Equal blocks along the order of falling degree: for graphs without locality (power-law), where the only
    // structure worth having is that the hubs sit together (er_panel.cpp: fewer partial sums).
    auto degree_blocks = [&]() -> int {
        std::vector<int> ord;
        degree_order(n, xadj, &ord);
        if (by_degree) *by_degree = true;
        return contiguous(ord.data());
    };
    if (cfg.partitioner == EHYB_PART_DEGREE) return degree_blocks();

    // ---- coarsening
    const double t0 = wall_seconds();
    const int coarse_target = std::max(nparts * 24, 1024);
    const int max_vw = (int)std::max<int64_t>(max_v, total / ((int64_t)nparts * 8));
    std::vector<Graph> levels;
    std::vector<std::vector<int>> cmaps;
    GView cur = fine;
    while (cur.n > coarse_target) {
        Graph cg;
        std::vector<int> cmap;
        coarsen_once(cur, max_vw, rng, &cg, &cmap);
        if (cg.n > cur.n * 0.93) {  // matching stalled
            // EHYB_PART_AUTO: a graph that stops coarsening before it has even halved, far from the target, has
            // no locality for a k-way partitioner to find (R-MAT 2^24: 124 M of 133 M edges cut after
            // 110 s, most of them in the initial partition of a 15 M-vertex "coarsest" graph).
            // Blocks of the degree order cost nothing and cut about as much (round 2: contiguous blocks of the
            // given numbering; the degree order leaves the panel-form residual 35-50 % fewer partial sums).
            if (cfg.partitioner == EHYB_PART_AUTO && cg.n > 8 * (int64_t)coarse_target && cg.n > n / 2) {
                if (cfg.verbose) printf("partition: matching stalled at %d of %d vertices: blocks of the degree order\n", cg.n, n);
                return degree_blocks();
            }
            if (cg.n < cur.n) {
                levels.push_back(std::move(cg));
                cmaps.push_back(std::move(cmap));
            }
            break;
        }
        levels.push_back(std::move(cg));
        cmaps.push_back(std::move(cmap));
        cur = levels.back().view();
    }
    if (!levels.empty()) cur = levels.back().view();
    const double t1 = wall_seconds();

    // ---- initial partition on the coarsest graph (cap relaxed by the largest vertex)
    int coarse_max_v = 1;
    for (int v = 0; v < cur.n; ++v) coarse_max_v = std::max(coarse_max_v, cur.wv(v));
    std::vector<int> cpart;
    std::vector<int64_t> pw;
    int64_t avg = (total + nparts - 1) / nparts;
    int64_t level_cap = std::max(cap, avg + coarse_max_v);
    initial_partition(cur, nparts, level_cap, rng, &cpart, &pw);
    const double t2 = wall_seconds();

    // ---- uncoarsening + refinement
    for (int l = (int)levels.size() - 1; l >= 0; --l) {
        GView finer = l == 0 ? fine : levels[l - 1].view();
        std::vector<int> fpart(finer.n);
        const std::vector<int>& cmap = cmaps[l];
        for (int v = 0; v < finer.n; ++v) fpart[v] = cpart[cmap[v]];
        cpart.swap(fpart);
        int lvl_max = 1;
        if (finer.vw)
            for (int v = 0; v < finer.n; ++v) lvl_max = std::max(lvl_max, finer.vw[v]);
        int64_t lcap = l == 0 ? cap : std::max(cap, avg + lvl_max);
        const double tl = wall_seconds();
        const int mv = refine_kway(finer, nparts, lcap, l == 0 ? 4 : 6, cpart, pw, rng);
        if (cfg.verbose > 1) printf("  refine level %d: n=%d moves %d %.3fs\n", l, finer.n, mv, wall_seconds() - tl);
    }
    if (levels.empty()) refine_kway(fine, nparts, cap, 4, cpart, pw, rng);
    enforce_cap(fine, nparts, cap, cpart, pw);
    // a last polish under the hard cap
    refine_kway(fine, nparts, cap, 2, cpart, pw, rng);
    const double t3 = wall_seconds();

    for (int p = 0; p < nparts; ++p)
        if (pw[p] > cap)
            EHYB_FAIL(EHYB_ERR_INTERNAL, "partition_graph: part %d has weight %lld > cap %lld", p,
                      (long long)pw[p], (long long)cap);
    std::copy(cpart.begin(), cpart.end(), part);
    int64_t cut = edge_cut(fine, part);
    if (edgecut) *edgecut = cut;
    if (cfg.verbose)
        printf("partition: n=%d parts=%d levels=%zu cut=%lld  coarsen %.2fs init %.2fs refine %.2fs\n", n,
               nparts, levels.size(), (long long)cut, t1 - t0, t2 - t1, t3 - t2);
    return EHYB_OK;
}

int refine_partition(int n, const int64_t* xadj, const int* adjncy, const int* vwgt, int nparts, int max_part_w, const Config& cfg, int* part,
                     int64_t* edgecut)
{
    GView fine;
    fine.n = n;
    fine.xadj = xadj;
    fine.adj = adjncy;
    fine.ew = nullptr;
    fine.vw = vwgt;
    std::vector<int> p(part, part + n);
    std::vector<int64_t> pw((size_t)nparts, 0);
    for (int v = 0; v < n; ++v) pw[(size_t)p[v]] += fine.wv(v);
    uint64_t rng = 0x1234ABCDull + (uint64_t)cfg.seed * 0x9E3779B97F4A7C15ull + 77;
    const int64_t cap = max_part_w;
    const double t0 = wall_seconds();
    const int mv = refine_kway(fine, nparts, cap, 4, p, pw, rng);
    enforce_cap(fine, nparts, cap, p, pw);
    refine_kway(fine, nparts, cap, 2, p, pw, rng);
    for (int q = 0; q < nparts; ++q)
        if (pw[(size_t)q] > cap) EHYB_FAIL(EHYB_ERR_INTERNAL, "refine_partition: part %d has weight %lld > cap %lld", q, (long long)pw[(size_t)q], (long long)cap);
    std::copy(p.begin(), p.end(), part);
    const int64_t cut = edge_cut(fine, part);
    if (edgecut) *edgecut = cut;
    if (cfg.verbose) printf("refinement on the rows themselves: %d moves, cut %lld, %.2fs\n", mv, (long long)cut, wall_seconds() - t0);
    return EHYB_OK;
}

}  // namespace ehyb

extern "C" int ehyb_partition_graph(int n, const int64_t* xadj, const int* adjncy, const int* vwgt,
                                    int nparts, int max_part_rows, const ehyb_config* cfg, int* part,
                                    int64_t* edgecut)
{
    ehyb::Config c = ehyb::resolve_config(cfg);
    ehyb::OmpScope omp_scope(c.host_threads);
    return ehyb::partition_graph(n, xadj, adjncy, vwgt, nparts, max_part_rows, c, part, edgecut);
}
