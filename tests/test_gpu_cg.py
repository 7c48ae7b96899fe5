"""The iterative caller: device-resident CG on the plan API, where the x of
every multiply differs -- the case in which the reference's "residual computed once" defect
(spmv.cu:41 vs kernel.cu:171-176) would give wrong answers."""
import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spla

pytestmark = pytest.mark.gpu


def spd_matrix(nx, ny, extra, seed):
    """2-D 5-point Laplacian + a few random symmetric couplings, made strictly diagonally dominant."""
    rng = np.random.default_rng(seed)
    n = nx * ny
    idx = np.arange(n).reshape(ny, nx)
    r = np.concatenate([idx[:, :-1].ravel(), idx[:-1, :].ravel(), rng.integers(0, n, extra)])
    c = np.concatenate([idx[:, 1:].ravel(), idx[1:, :].ravel(), rng.integers(0, n, extra)])
    keep = r != c
    r, c = r[keep], c[keep]
    v = -rng.uniform(0.5, 1.5, len(r))
    A = sp.coo_matrix((np.concatenate([v, v]), (np.concatenate([r, c]), np.concatenate([c, r]))), shape=(n, n)).tocsr()
    A.sum_duplicates()
    d = np.asarray(abs(A).sum(axis=1)).ravel() + 0.05
    return (A + sp.diags(d)).tocsr()


def cpu_cg(A, b, max_iter, rtol):
    """The same recurrences on the CPU (the checker), iteration for iteration."""
    x = np.zeros_like(b)
    r = b.copy()
    p = r.copy()
    rs = r @ r
    bb = b @ b
    hist = [np.sqrt(rs / bb)]
    it = 0
    while it < max_iter and np.sqrt(rs / bb) > rtol:
        q = A @ p
        alpha = rs / (p @ q)
        x += alpha * p
        r -= alpha * q
        rs_new = r @ r
        p = r + (rs_new / rs) * p
        rs = rs_new
        it += 1
        hist.append(np.sqrt(rs / bb))
    return x, it, hist


@pytest.mark.parametrize("mode,lds,sym", [(2, 2048, 0), (1, 512, 0), (2, 2048, 1)], ids=["halo", "refwindow", "symmetric-pairs"])
def test_cg_matches_cpu_and_scipy(E, O, gpu, mode, lds, sym):
    """The third arm runs CG on symmetric pair storage -- an SPD matrix is symmetric, so the solver's
    multiply reads every in-partition pair once."""
    A = spd_matrix(120, 100, 3000, 1)
    n = A.shape[0]
    cfg = E.make_config(window_mode=mode, lds_doubles=lds, sym_pairs=sym)
    m = E.Matrix.from_csr(A.indptr, A.indices, A.data, cfg, symmetric=True)
    m.reorder(cfg)
    perm = m.reorder_list.copy()
    plan = E.Plan(m, cfg)
    if mode == 1:
        assert plan.stats["nnz_er"] > 0, "this arm must exercise the residual on every iteration"
    if sym:
        assert plan.stats["sym_pairs"] > 0.25 * A.nnz
    b = O.x_glibc(n) + 0.3
    xp, iters, rel = plan.cg(E.vector_reorder(b, perm), max_iter=400, rtol=1e-10, check_every=1)
    x = E.vector_recover(xp, perm)
    x_cpu, it_cpu, hist = cpu_cg(A, b, 400, 1e-10)
    assert rel <= 1e-10 and abs(iters - it_cpu) <= 2, (iters, it_cpu, rel)
    assert np.linalg.norm(A @ x - b) <= 2e-10 * np.linalg.norm(b)
    assert np.linalg.norm(x - x_cpu) <= 1e-8 * np.linalg.norm(x_cpu)
    x_sp, info = spla.cg(A, b, rtol=1e-12, maxiter=2000)
    assert info == 0 and np.linalg.norm(x - x_sp) <= 1e-7 * np.linalg.norm(x_sp)


def test_jacobi_pcg_on_a_badly_scaled_system(E, O, gpu):
    """ehyb_pcg: the diagonal preconditioner (the reference's PRECOND switch).  D A D with a widely
    varying D is SPD but badly scaled: plain CG needs many times the iterations of Jacobi-PCG, and
    both reach the same solution."""
    A0 = spd_matrix(100, 90, 2000, 3)
    n = A0.shape[0]
    d = 10.0 ** np.random.default_rng(5).uniform(-2, 2, n)
    A = (sp.diags(d) @ A0 @ sp.diags(d)).tocsr()
    cfg = E.make_config(lds_doubles=2048, sym_pairs=1)
    m = E.Matrix.from_csr(A.indptr, A.indices, A.data, cfg, symmetric=True)
    m.reorder(cfg)
    perm = m.reorder_list.copy()
    plan = E.Plan(m, cfg)
    b = A @ np.ones(n)
    bp = E.vector_reorder(b, perm)
    inv_diag = E.vector_reorder(1.0 / A.diagonal(), perm)
    xp, it_pcg, rel = plan.cg(bp, max_iter=3000, rtol=1e-9, check_every=5, inv_diag=inv_diag)
    x = E.vector_recover(xp, perm)
    assert rel <= 1e-9 and np.linalg.norm(A @ x - b) <= 5e-9 * np.linalg.norm(b)
    _, it_cg, rel_cg = plan.cg(bp, max_iter=3000, rtol=1e-9, check_every=5)
    assert it_pcg * 3 < it_cg or rel_cg > 1e-9, (it_pcg, it_cg, rel_cg)
    # the same recurrences on the CPU stop within a few iterations of the device
    Minv = sp.diags(1.0 / A.diagonal())
    xc, rc = np.zeros(n), b.copy()
    zc = Minv @ rc
    pc, rz, k = zc.copy(), rc @ zc, 0
    while np.linalg.norm(rc) > 1e-9 * np.linalg.norm(b) and k < 3000:
        q = A @ pc
        alpha = rz / (pc @ q)
        xc += alpha * pc
        rc -= alpha * q
        zc = Minv @ rc
        rz, rz_old = rc @ zc, rz
        pc = zc + (rz / rz_old) * pc
        k += 1
    assert abs(k - it_pcg) <= 5 + 0.05 * k, (k, it_pcg)


def test_cg_stops_at_max_iter_and_reports_breakdown(E, O, gpu):
    A = spd_matrix(60, 50, 500, 2)
    cfg = E.make_config(lds_doubles=1024)
    m = E.Matrix.from_csr(A.indptr, A.indices, A.data, cfg, symmetric=True)
    m.reorder(cfg)
    plan = E.Plan(m, cfg)
    b = E.vector_reorder(np.ones(A.shape[0]), m.reorder_list)
    _, iters, rel = plan.cg(b, max_iter=7, rtol=1e-30, check_every=3)
    assert iters == 7 and 0 < rel < 1
    # an indefinite matrix breaks CG down: reported as an error, not a silent NaN
    B = (sp.diags(np.r_[np.ones(A.shape[0] // 2), -np.ones(A.shape[0] - A.shape[0] // 2)]) @ A).tocsr()
    B = ((B + B.T) * 0.5).tocsr()
    mb = E.Matrix.from_csr(B.indptr, B.indices, B.data, cfg, symmetric=True)
    mb.reorder(cfg)
    pb = E.Plan(mb, cfg)
    try:
        _, it2, rel2 = pb.cg(E.vector_reorder(np.ones(A.shape[0]), mb.reorder_list), max_iter=200, rtol=1e-12)
        assert not np.isnan(rel2)
    except E.EhybError as e:
        assert "breakdown" in str(e)


def test_cg_is_reproducible_and_graph_replay_changes_nothing(E, O, gpu, monkeypatch):
    """The dot products are summed in a fixed order (per-workgroup partials, no atomics), so with
    plain storage two solves agree bit for bit -- and so does a solve issued launch by launch
    (cfg.graphs = 2) instead of replaying the captured pair of iterations."""
    A = spd_matrix(110, 90, 2500, 4)
    cfg = E.make_config(lds_doubles=2048, sym_pairs=0)
    m = E.Matrix.from_csr(A.indptr, A.indices, A.data, cfg, symmetric=True)
    m.reorder(cfg)
    plan = E.Plan(m, cfg)
    b = E.vector_reorder(O.x_glibc(A.shape[0]) + 0.1, m.reorder_list)
    x1, it1, rel1 = plan.cg(b, max_iter=41, rtol=1e-30, check_every=8)
    x2, it2, rel2 = plan.cg(b, max_iter=41, rtol=1e-30, check_every=8)
    plain = E.Plan(m, E.make_config(lds_doubles=2048, sym_pairs=0, graphs=2))
    x3, it3, rel3 = plain.cg(b, max_iter=41, rtol=1e-30, check_every=8)
    assert it1 == it2 == it3 == 41
    assert np.array_equal(x1, x2) and rel1 == rel2
    assert np.array_equal(x1, x3) and rel1 == rel3


@pytest.mark.parametrize("sym", [0, 1], ids=["plain", "symmetric-pairs"])
def test_dot_product_left_by_the_multiply(E, O, gpu, sym):
    """p . (A p) as a by-product of the multiply (one partial per workgroup of the window launch, cfg.cg_fused_dot) against the
    separate dot kernel (cfg.cg_fused_dot = 2): the same solve to the rounding of the two summation orders, both at the CPU's
    iteration count; a matrix whose plan has a residual launch of its own keeps the dot kernel (nothing to compare: same code)."""
    A = spd_matrix(120, 100, 3000, 7)
    n = A.shape[0]
    kw = dict(lds_doubles=2048, sym_pairs=sym, direct=2)
    cfg = E.make_config(**kw)
    m = E.Matrix.from_csr(A.indptr, A.indices, A.data, cfg, symmetric=True)
    m.reorder(cfg)
    perm = m.reorder_list.copy()
    fused, separate = E.Plan(m, cfg), E.Plan(m, E.make_config(cg_fused_dot=2, **kw))
    st = fused.stats
    assert st["nnz_er"] == 0 or st["er_inline"] > 0, "this matrix must multiply in one window launch"
    b = E.vector_reorder(O.x_glibc(n) + 0.3, perm)
    x1, it1, rel1 = fused.cg(b, max_iter=400, rtol=1e-10, check_every=2)
    x2, it2, rel2 = separate.cg(b, max_iter=400, rtol=1e-10, check_every=2)
    _, it_cpu, _ = cpu_cg(A, O.x_glibc(n) + 0.3, 400, 1e-10)
    assert rel1 <= 1e-10 and rel2 <= 1e-10 and abs(it1 - it_cpu) <= 2 and abs(it2 - it_cpu) <= 2, (it1, it2, it_cpu)
    assert np.linalg.norm(x1 - x2) <= 1e-9 * np.linalg.norm(x2)
    x = E.vector_recover(x1, perm)
    assert np.linalg.norm(A @ x - (O.x_glibc(n) + 0.3)) <= 2e-10 * np.linalg.norm(O.x_glibc(n) + 0.3)
    # fixed summation order in both: a second solve repeats the first bit for bit on plain storage
    if not sym:
        x1b, _, rel1b = fused.cg(b, max_iter=400, rtol=1e-10, check_every=2)
        assert np.array_equal(x1, x1b) and rel1 == rel1b
