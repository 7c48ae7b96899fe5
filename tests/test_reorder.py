"""Host pre-step: partitioner and matrixReorder equivalents (reference reordering.c), CPU only."""
import numpy as np
import pytest
import scipy.sparse as sp

from oracle import ehyb_ref_layout as R


def grid_graph(nx, ny):
    idx = np.arange(nx * ny).reshape(ny, nx)
    e = [(idx[:, :-1].ravel(), idx[:, 1:].ravel()), (idx[:-1, :].ravel(), idx[1:, :].ravel())]
    r = np.concatenate([a for a, _ in e] + [b for _, b in e])
    c = np.concatenate([b for _, b in e] + [a for a, _ in e])
    A = sp.coo_matrix((np.ones(len(r)), (r, c)), shape=(nx * ny, nx * ny)).tocsr()
    return A


def cut_of(A, part):
    C = A.tocoo()
    return int(np.count_nonzero(part[C.row] != part[C.col]) // 2)


@pytest.mark.parametrize("nparts", [2, 7, 16, 64])
def test_partition_is_valid_and_compact(E, nparts):
    A = grid_graph(96, 80)
    n = A.shape[0]
    cap = int(np.ceil(n / nparts * 1.03))
    part, cut = E.partition_graph(A.indptr, A.indices, nparts, cap)
    assert part.min() >= 0 and part.max() < nparts
    sizes = np.bincount(part, minlength=nparts)
    assert sizes.max() <= cap and sizes.sum() == n
    assert cut == cut_of(A, part)
    # compact parts: within 1.5x of a square tiling (perimeter 4*sqrt(area) per part, every
    # cut edge shared by two parts)
    ideal = 4 * np.sqrt(n / nparts) * nparts / 2
    assert cut < 1.5 * ideal + 40, (cut, ideal)
    # deterministic for a given seed
    part2, cut2 = E.partition_graph(A.indptr, A.indices, nparts, cap)
    assert np.array_equal(part, part2) and cut == cut2


def test_partition_hard_cap_and_errors(E):
    A = grid_graph(40, 40)
    n = A.shape[0]
    part, _ = E.partition_graph(A.indptr, A.indices, 10, 160)  # exactly n/10: zero slack
    assert np.bincount(part).max() <= 160
    with pytest.raises(E.EhybError):
        E.partition_graph(A.indptr, A.indices, 10, 150)  # 10 x 150 < 1600
    one, cut = E.partition_graph(A.indptr, A.indices, 1)
    assert not one.any() and cut == 0
    # vertex weights (used for the per-GPU blocks): balance on weight, not count
    w = np.ones(n, dtype=np.int32)
    w[: n // 4] = 5
    part, _ = E.partition_graph(A.indptr, A.indices, 4, 0, vwgt=w)
    pw = np.bincount(part, weights=w, minlength=4)
    assert pw.max() <= w.sum() / 4 * 1.001 + 1 + 5


def test_partition_contiguous_and_disconnected(E):
    A = sp.block_diag([grid_graph(10, 10), grid_graph(7, 9), sp.csr_matrix((5, 5))]).tocsr()  # 5 isolated vertices
    n = A.shape[0]
    part, _ = E.partition_graph(A.indptr, A.indices, 6, 32)
    assert np.bincount(part, minlength=6).max() <= 32
    cfg = E.make_config(partitioner=E.EHYB_PART_CONTIGUOUS)
    part, _ = E.partition_graph(A.indptr, A.indices, 4, 64, cfg=cfg)
    assert np.all(np.diff(part) >= 0) and part[0] == 0 and part[-1] <= 3
    # unit weights: chunks are whole 64-row slabs, so block-structured inputs stay aligned
    assert np.all(np.bincount(part)[:-1] % 64 == 0)


@pytest.mark.parametrize("kind,args,sym", [
    ("stencil2d", (40, 30, 9, 300, 2), True),
    ("rmat", (10, 1 << 13, 4), False),
    ("fem3d", (3000, 3, 10, 10, 13500, 1, 5), True),
])
@pytest.mark.parametrize("mode", [1, 2])
def test_matrix_reorder_contract(E, O, kind, args, sym, mode):
    """What matrixReorder[_unsym] promises (reordering.c:231-378 / 41-228)."""
    cfg = E.make_config(window_mode=mode, lds_doubles=256)
    m = E.Matrix.generate(kind, *args, cfg=cfg)
    A = m.to_scipy()
    n, nnz = m.n, m.nnz
    nparts, cache = m.c.nParts, int(m.c.vectorCacheSize)
    m.reorder(cfg, symmetric=sym)
    lst = m.reorder_list.copy()
    assert sorted(lst.tolist()) == list(range(n)), "reorderList is a permutation"
    pb = m.part_boundary
    assert pb[0] == 0 and pb[-1] == n and len(pb) == nparts + 1
    assert np.diff(pb).max() <= cache
    assert m.nnz == nnz
    rp, I, J, V = m.row_idx, m.I, m.J, m.V
    assert np.array_equal(np.repeat(np.arange(n), np.diff(rp)), I), "row-grouped (rowIdx consistent with I)"
    assert np.array_equal(m.num_in_row, np.diff(rp))
    # P A P^T: entry (i,j) of A sits at (list[i], list[j])
    B = sp.csr_matrix((V, J, rp.astype(np.int64)), shape=(n, n))
    inv = np.empty(n, dtype=np.int64)
    inv[lst] = np.arange(n)
    assert abs(B[lst][:, lst] - A).max() == 0 if False else abs(B - A[inv][:, inv]).max() == 0
    # entries of a row keep their original relative order (reordering.c:348-357)
    i_old = int(np.argmax(np.diff(A.indptr)))
    new_row = lst[i_old]
    assert np.array_equal(J[rp[new_row]:rp[new_row + 1]], lst[A.indices[A.indptr[i_old]:A.indptr[i_old + 1]]])
    # numInRow2 = entries inside [partStart, partStart + cache) (reordering.c:358-361)
    assert np.array_equal(m.num_in_row2, R.num_in_row2(rp, J, pb, cache))
    # rows of a partition are sorted by the ELL-entry key, descending (reordering.c:334):
    # in reference-window mode the key is the in-partition count
    if mode == 1:
        part_of = np.searchsorted(pb, np.arange(n), side="right") - 1
        key = np.zeros(n, dtype=np.int64)
        np.add.at(key, I[part_of[I] == part_of[J]], 1)
        for p in range(nparts):
            k = key[pb[p]:pb[p + 1]]
            assert np.all(np.diff(k) <= 0), f"partition {p} not sorted by in-partition entries"
    # vectorReorder / vectorRecover round trip (reordering.c:380-391)
    x = O.x_glibc(n)
    xp = E.vector_reorder(x, lst)
    assert np.array_equal(xp[lst], x) and np.array_equal(E.vector_recover(xp, lst), x)
    # the permuted product, un-permuted, is the original product
    y = E.vector_recover(B @ xp, lst)
    assert np.allclose(y, A @ x, rtol=0, atol=1e-13)


def test_reorder_rejects_bad_input(E):
    cfg = E.make_config()
    m = E.Matrix.generate("stencil2d", 10, 10, 5, 0, 1, cfg=cfg)
    m.c.nParts = 0
    with pytest.raises(E.EhybError):
        m.reorder(cfg)
    m.c.nParts = 2
    m.J[3] = 1000
    with pytest.raises(E.EhybError):
        m.reorder(cfg)


def test_sizing(E):
    """The MI355X re-derivation of solver_test.c:158-182: window from the LDS budget, parts from it."""
    cfg = E.make_config(window_mode=2, lds_doubles=10240)
    nparts, cache, kpp = E.sizing(943695, cfg)
    assert cache == 5632 and cache <= 10240
    assert nparts * cache >= 943695 and (nparts - 1) * cache < 943695 * 1.04
    assert kpp >= 1
    ref = E.make_config(window_mode=1, lds_doubles=20480)
    nparts, cache, _ = E.sizing(4194304, ref)           # config 3: past the reference's int16 limit
    assert cache == 20480 and nparts * cache >= 4194304
    small = E.sizing(10974, E.make_config())            # bcsstk17-sized
    assert small[0] >= 1 and small[1] <= 20480
    with pytest.raises(E.EhybError):
        E.sizing(0)


def test_partition_does_not_depend_on_the_thread_count(E):
    """Large graphs are matched in parallel rounds of mutual proposals before the greedy pass, and
    contracted in parallel: both are written so that the outcome is the same for any number of
    threads (40,000 vertices: above the size where the parallel rounds start)."""
    from threadpoolctl import threadpool_limits

    A = grid_graph(250, 160)
    n = A.shape[0]
    cap = int(np.ceil(n / 48 * 1.03))
    part_many, cut_many = E.partition_graph(A.indptr, A.indices, 48, cap)
    with threadpool_limits(limits=1, user_api="openmp"):
        part_one, cut_one = E.partition_graph(A.indptr, A.indices, 48, cap)
    with threadpool_limits(limits=3, user_api="openmp"):
        part_three, _ = E.partition_graph(A.indptr, A.indices, 48, cap)
    assert np.array_equal(part_many, part_one) and np.array_equal(part_many, part_three) and cut_many == cut_one
    assert np.bincount(part_many, minlength=48).max() <= cap
    ideal = 4 * np.sqrt(n / 48) * 48 / 2
    assert cut_many < 1.5 * ideal + 40, (cut_many, ideal)


def test_unknowns_of_a_node_stay_together(E):
    """The k-way partition runs on the compressed graph where rows share their column lists (reorder.cpp,
    partition_compressed): the three unknowns of a finite-element node -- rows with one column list -- are one
    vertex there, so no partition separates them; sizes stay under the cap; deterministic."""
    cfg = E.make_config(lds_doubles=4096, partitioner=E.EHYB_PART_MULTILEVEL, graph_compress=1)  # (automatic: with symmetric pairs only)
    m = E.Matrix.generate("fem3d", 30000, 3, 22, 22, 13500, 1, 1, cfg=cfg)
    rp, J = m.row_idx.astype(np.int64).copy(), m.J.copy()
    n = m.n
    twin = np.zeros(n, dtype=bool)       # row has the column list of the row above
    for r in range(1, n):
        a, b, c = rp[r - 1], rp[r], rp[r + 1]
        twin[r] = (c - b == b - a) and np.array_equal(J[a:b], J[b:c])
    assert twin.mean() > 0.6             # 3 unknowns per node: two of three rows
    m.reorder(cfg)
    lst = m.reorder_list
    pb = m.part_boundary[:m.c.nParts + 1]
    assert pb[0] == 0 and pb[-1] == n and np.all(np.diff(pb) >= 0)
    part_of_new = np.searchsorted(pb, np.arange(n), side="right") - 1
    part_of_old = part_of_new[lst]
    assert np.all(part_of_old[1:][twin[1:]] == part_of_old[:-1][twin[1:]]), "a node's unknowns were separated"
    assert np.diff(pb).max() <= max(int(m.c.vectorCacheSize), -(-n // m.c.nParts))
    m2 = E.Matrix.generate("fem3d", 30000, 3, 22, 22, 13500, 1, 1, cfg=cfg)
    m2.reorder(cfg)
    assert np.array_equal(m2.reorder_list, lst)


def test_compressed_graph_with_a_refinement_on_the_rows(E, O):
    """cfg.graph_compress = 3: the compressed graph is the first coarsening level only -- the partition it gives is refined once
    on the rows themselves, where the unknowns of a node may part.  Partitions under the cap, every row in one, deterministic, and
    the plan built on them multiplies like the oracle."""
    from util import Case
    cfg = E.make_config(lds_doubles=4096, partitioner=E.EHYB_PART_MULTILEVEL, graph_compress=3)
    c = Case(E, O, "fem3d", (30000, 3, 22, 22, 13500, 1, 1), cfg)
    m = c.m
    n = m.n
    pb = m.part_boundary[:m.c.nParts + 1]
    assert pb[0] == 0 and pb[-1] == n and np.all(np.diff(pb) >= 0)
    assert np.diff(pb).max() <= max(int(m.c.vectorCacheSize), -(-n // m.c.nParts))
    assert np.array_equal(np.sort(m.reorder_list), np.arange(n))
    plan = E.Plan(m, cfg, upload=False)
    yp, written = O.walk_plan(plan, c.xp)
    bad, worst = c.check(yp)
    assert bad == 0, worst
    m2 = E.Matrix.generate("fem3d", 30000, 3, 22, 22, 13500, 1, 1, cfg=cfg)
    m2.reorder(cfg)
    assert np.array_equal(m2.reorder_list, c.perm)


def test_graded_mesh_asks_for_fewer_partitions(E, O):
    """Symmetric pair storage on a matrix whose rows differ a lot in length: entry-balanced partitions, the
    row-limited ones bisected -- and fewer asked for up front, so that the launch ends at one round of 256
    workgroups (reorder.cpp).  nParts is written back whichever way the count moved."""
    cfg = E.make_config(sym_pairs=1)
    m = E.Matrix.generate("fem3d_graded", 300000, 3, 46, 46, 100000, 705000, 1, 1, cfg=cfg)
    asked = int(m.c.nParts)
    assert asked == 256
    x = O.x_glibc(m.n)
    y_ref = O.spmv_coo(m.n, m.I, m.J, m.V, x)
    scale = O.abs_rowsum(m.n, m.I, m.J, m.V, x)
    m.reorder(cfg)
    got = int(m.c.nParts)
    pb = m.part_boundary[:got + 1]
    assert 200 <= got <= 256, got
    assert pb[0] == 0 and pb[-1] == m.n and np.all(np.diff(pb) >= 0)
    plan = E.Plan(m, cfg, upload=False)
    st = plan.stats
    assert st["n_items"] <= 256 and st["sym_pairs"] > 0
    y, written = O.walk_plan(plan, E.vector_reorder(x, m.reorder_list))
    assert written[:m.n].min() == 1 and written[:m.n].max() == 1
    bad, worst = O.check_tolerance(E.vector_recover(y, m.reorder_list), y_ref, scale)
    assert bad == 0, worst


def test_append_rows_gives_a_rank_its_foreign_rows(E):
    """ehyb_matrix_append_rows (exchange "cover"): rows behind the last partition receive entries in the rank's own columns and
    become partitions of their own; what it refuses."""
    import ctypes as C

    import numpy as np

    from ehyb_spmv_gpu_amd import _lib

    lib = _lib.load()
    cfg = E.make_config(lds_doubles=1024)
    m = E.Matrix.generate("stencil2d", 40, 40, 5, 200, 3, cfg=cfg)
    m.reorder(cfg)
    n0, nnz0, np0 = m.n, m.nnz, int(m.c.nParts)
    assert int(m.part_boundary[np0]) == n0
    m.append_ghosts(700, np.array([0, 1, 5], dtype=np.int32), np.array([3, 0, 699], dtype=np.int32), np.array([1.0, 2.0, 3.0]))
    assert m.n == n0 + 700 and m.nnz == nnz0 + 3

    def call(row0, n_rows, ri, cj, v, per=0):
        ri, cj, v = (np.ascontiguousarray(a, dtype=t) for a, t in ((ri, np.int32), (cj, np.int32), (v, np.float64)))
        return lib.ehyb_matrix_append_rows(C.byref(m.c), row0, n_rows, len(ri), ri.ctypes.data_as(C.POINTER(C.c_int)), cj.ctypes.data_as(C.POINTER(C.c_int)),
                                           v.ctypes.data_as(C.POINTER(C.c_double)), per)

    assert call(n0 - 1, 5, [0], [0], [1.0]) == 1 and b"inside the partitions" in lib.ehyb_last_error()
    assert call(n0, 701, [0], [0], [1.0]) == 1                                    # beyond the dimension
    assert call(n0, 10, [3, 2], [0, 0], [1.0, 1.0]) == 1                          # rows not ascending
    assert call(n0, 10, [0], [m.n], [1.0]) == 1                                   # column out of range
    assert m.nnz == nnz0 + 3 and int(m.c.nParts) == np0                           # nothing changed
    ri = np.repeat(np.arange(600, dtype=np.int32), 2)
    cj = (np.arange(1200, dtype=np.int32) * 7) % n0
    assert call(n0, 600, ri, cj, np.ones(1200), 256) == 0
    m.refresh() if hasattr(m, "refresh") else None
    assert int(m.c.totalNum) == nnz0 + 3 + 1200 and int(m.c.nParts) == np0 + 3   # 256 + 256 + 88 rows
    pb = np.ctypeslib.as_array(m.c.partBoundary, shape=(int(m.c.nParts) + 1,))
    assert list(pb[np0:]) == [n0, n0 + 256, n0 + 512, n0 + 600]
    rp = np.ctypeslib.as_array(m.c.rowIdx, shape=(m.n + 1,))
    assert rp[n0 + 600] - rp[n0] == 1200 and rp[m.n] == nnz0 + 3 + 1200 and np.all(np.diff(rp[n0:n0 + 601]) == 2)
    # a plan over own + foreign rows, row blocks split at the boundary, walked by the oracle
    from oracle import oracle as O

    plan = E.Plan(m, E.make_config(lds_doubles=1024, n_top=2, er_mode=2, prune_pct=1, fuse_er=2, direct=2, row_split=n0, er_panel_cols=256, er_block_rows=100),
                  rows=(0, n0 + 600), upload=False)
    st = plan.stats
    assert st["nnz_ell"] == 0 and st["n_rows"] == n0 + 600 and st["er_partials"] > 0
    u2 = plan.array("pb_units2").reshape(-1, 4)
    assert not np.any((u2[:, 2] < n0) & (u2[:, 2] + np.abs(u2[:, 3]) > n0))
    x = np.linspace(-1.0, 1.0, m.n)
    y, w = O.walk_plan(plan, x)
    J = np.ctypeslib.as_array(m.c.J, shape=(int(m.c.totalNum),))
    V = np.ctypeslib.as_array(m.c.V, shape=(int(m.c.totalNum),))
    want = np.add.reduceat(V * x[J], rp[:n0 + 600].astype(np.int64))
    want[np.diff(rp[:n0 + 601]) == 0] = 0.0
    assert np.allclose(np.asarray(y)[:n0 + 600], want, rtol=1e-13, atol=1e-13)
