"""GPU parity: the HIP path, called through the C-ABI, against the CPU oracle.

Tolerance (fp64, stated in SURVEY.md 8c / BASELINE.md 4): per row
    |y_gpu - y_cpu| <= 1e-12 * sum_j |a_ij * x_j|
EHYB changes the summation order (ELL pairs, then += residual), so bit equality with the
row-ordered CPU sum is not expected; the reference's own check is 1 % relative
(solver_test.c:389) and is also evaluated.
"""
import numpy as np
import pytest

from util import SMALL_CASES, Case, fem_plus_rmat

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,kind,args", SMALL_CASES, ids=[c[0] for c in SMALL_CASES])
@pytest.mark.parametrize("mode", [1, 2], ids=["refwindow", "halo"])
def test_spmvGPuEHYB_matches_oracle(E, O, gpu, name, kind, args, mode):
    """The drop-in symbol end to end: reorder -> spmvGPuEHYB -> recover -> compare."""
    cfg = E.make_config(window_mode=mode, lds_doubles=4096)
    c = Case(E, O, kind, args, cfg)
    yp, iters = E.spmv_gpu_ehyb(c.m, c.xp, 3)
    assert iters == 3
    bad, worst = c.check(yp)
    assert bad == 0, f"{name}: {bad} rows over tolerance, worst {worst:.3e}"
    # The reference's own check (solver_test.c:389): 1 % of min(|a|,|b|).  It has no pass/fail
    # and trips on rows whose terms cancel to ~0, so only rows that are NOT cancellation rows
    # may be flagged.
    y = c.recover(yp)
    loose_bad, diff, ampl = O.compare(y, c.y_ref, 0.01)
    cancel = int(np.count_nonzero(np.abs(c.y_ref) < 1e-10 * c.scale))
    assert loose_bad <= cancel
    assert diff <= 1e-12 * c.scale.sum()


@pytest.mark.parametrize("threads", [256, 512, 1024])
@pytest.mark.parametrize("lds", [1024, 6144, 20480])
def test_plan_configs(E, O, gpu, threads, lds):
    """Workgroup sizes and window sizes up to the full 160 KiB of LDS."""
    cfg = E.make_config(window_mode=2, lds_doubles=lds, threads=threads, items_per_cu=2)
    c = Case(E, O, "fem3d", (30000, 3, 22, 22, 13500, 1, 1), cfg)
    plan = E.Plan(c.m, cfg)
    yp = plan.spmv_host(c.xp, iters=2)
    bad, worst = c.check(yp)
    assert bad == 0, f"threads={threads} lds={lds}: worst {worst:.3e}"
    st = plan.stats
    assert st["nnz_ell"] + st["nnz_er"] == c.nnz


def test_residual_recomputed_every_iteration(E, O, gpu):
    """The reference computes the residual only on its first launch (spmv.cu:41 vs
    kernel.cu:171-176); here x may change between multiplies."""
    cfg = E.make_config(window_mode=1, lds_doubles=1024)
    c = Case(E, O, "rmat", (13, 1 << 16, 9), cfg)
    plan = E.Plan(c.m, cfg)
    assert plan.stats["nnz_er"] > 0
    x2 = c.x[::-1].copy()
    y_ref2 = O.spmv_coo(c.n, c.m.I, c.m.J, c.m.V, E.vector_reorder(x2, c.perm))  # permuted matrix & x
    dx, dy = E.DeviceBuffer(c.n), E.DeviceBuffer(c.n)
    dx.upload(c.xp)
    plan.spmv(dx.ptr, dy.ptr)
    dx.upload(E.vector_reorder(x2, c.perm))
    plan.spmv(dx.ptr, dy.ptr)
    y2 = dy.download()
    scale = O.abs_rowsum(c.n, c.m.I, c.m.J, c.m.V, E.vector_reorder(x2, c.perm))
    bad, worst = O.check_tolerance(y2, y_ref2, scale)
    assert bad == 0, f"second multiply with a new x is wrong: worst {worst:.3e}"


def test_long_rows_split_with_atomics(E, O, gpu):
    """Rows far longer than er_seg_len (R-MAT hubs): the working form of longRowKernel."""
    cfg = E.make_config(window_mode=1, lds_doubles=256, er_seg_len=64)
    c = Case(E, O, "rmat", (11, 1 << 17, 3), cfg)
    plan = E.Plan(c.m, cfg)
    seg_row = plan.array("er_seg_row")
    assert (seg_row < 0).any(), "expected split rows"
    yp = plan.spmv_host(c.xp)
    bad, worst = c.check(yp)
    assert bad == 0, f"worst {worst:.3e}"


def test_empty_residual(E, O, gpu):
    """Pure ELL (config 3's shape): the reference exit(0)s here (convert.c:136-139)."""
    cfg = E.make_config(window_mode=1, lds_doubles=4096, part_rows=3072, partitioner=E.EHYB_PART_CONTIGUOUS)
    c = Case(E, O, "banded", (1 << 15, 32, 1024), cfg)
    plan = E.Plan(c.m, cfg)
    st = plan.stats
    assert st["nnz_er"] == 0 and st["ell_padding"] == 0
    bad, worst = c.check(plan.spmv_host(c.xp))
    assert bad == 0


def test_linearity_and_phases(E, O, gpu):
    """A(ax+bz) = aAx + bAz, and phase 1 + phase 2 == phase 0."""
    cfg = E.make_config(window_mode=2, lds_doubles=2048)
    c = Case(E, O, "stencil2d", (150, 150, 9, 8000, 4), cfg)
    plan = E.Plan(c.m, cfg)
    rng = np.random.default_rng(0)
    z = rng.standard_normal(c.n)
    ya, yz = plan.spmv_host(c.xp), plan.spmv_host(z)
    ycomb = plan.spmv_host(2.5 * c.xp - 0.75 * z)
    scale = np.abs(2.5 * ya) + np.abs(0.75 * yz) + 1e-300
    assert np.max(np.abs(ycomb - (2.5 * ya - 0.75 * yz)) / scale) < 1e-10
    dx, dy = E.DeviceBuffer(c.n), E.DeviceBuffer(c.n)
    dx.upload(c.xp)
    plan.spmv(dx.ptr, dy.ptr, phase=1)
    plan.spmv(dx.ptr, dy.ptr, phase=2)
    if plan.stats["er_inline"] == 0:
        assert np.array_equal(dy.download(), ya)
    else:
        # a tiny residual rides inside the ELL launch of the one-call form (summed with the ELL entries
        # before y is rounded): the two-phase result may differ in the last bits of those rows
        assert c.check(dy.download())[0] == 0


PANEL_CASES = [
    ("rmat_s14", "rmat", (14, 1 << 17, 1), dict(lds_doubles=512, er_panel_cols=512, er_block_rows=300)),
    ("rmat_s12_dense", "rmat", (12, 1 << 18, 5), dict(lds_doubles=256, er_panel_cols=256, er_block_rows=64)),
    ("rmat_s17_defaults", "rmat", (17, 1 << 20, 3), dict()),
    ("kkt_contiguous", "kkt3d", (24,), dict(lds_doubles=2048, partitioner=1)),
    ("fem_reference_window", "fem3d", (30000, 3, 22, 22, 13500, 1, 1), dict(window_mode=1, lds_doubles=1024, er_panel_cols=16384)),
    ("rmat_sym_storage", "rmat", (14, 1 << 17, 4), dict(lds_doubles=512, sym_pairs=1, er_panel_cols=256)),
]


@pytest.mark.parametrize("name,kind,args,kw", PANEL_CASES, ids=[c[0] for c in PANEL_CASES])
def test_panel_form_of_the_residual(E, O, gpu, name, kind, args, kw):
    """er_mode = 2: the residual as two streaming launches (x panels, then y blocks in LDS) instead of
    gathers from global memory (csrc/er_panel.cpp) -- against the oracle, against the CSR form of the
    same plan data, with a changing x, and through the two-phase call the multi-GPU path uses."""
    cfg = E.make_config(er_mode=2, fuse_er=2, **kw)
    c = Case(E, O, kind, args, cfg)
    plan = E.Plan(c.m, cfg)
    st = plan.stats
    assert st["er_partials"] > 0
    y = plan.spmv_host(c.xp)
    bad, worst = c.check(y)
    assert bad == 0, f"{name}: {bad} rows over tolerance, worst {worst:.3e}"
    # the CSR form of the same matrix agrees to tolerance (different summation order)
    cfg1 = E.make_config(er_mode=1, fuse_er=2, **kw)
    y1 = E.Plan(c.m, cfg1).spmv_host(c.xp)
    assert c.check(y1)[0] == 0
    # the residual is recomputed for a new x (partial buffer fully rewritten), and phases add up
    x2 = E.vector_reorder(c.x[::-1].copy(), c.perm)
    dx, dy = E.DeviceBuffer(c.n).upload(c.xp), E.DeviceBuffer(c.n)
    plan.spmv(dx.ptr, dy.ptr)
    dx.upload(x2)
    plan.spmv(dx.ptr, dy.ptr, phase=1)
    plan.spmv(dx.ptr, dy.ptr, phase=2)
    y2 = dy.download()
    ref2 = O.spmv_coo(c.n, c.m.I, c.m.J, c.m.V, x2)
    sc2 = O.abs_rowsum(c.n, c.m.I, c.m.J, c.m.V, x2)
    assert O.check_tolerance(y2, ref2, sc2)[0] == 0
    # deterministic: no global atomics in either pass; LDS adds of one row block may reorder
    y_b = plan.spmv_host(c.xp)
    assert c.check(y_b)[0] == 0


DIRECT_CASES = [
    ("bcsstk17_like", "fem3d", (10974, 3, 62, 59, 250000, 1, 17)),
    ("rmat_s13_hubs", "rmat", (13, 1 << 18, 3)),
    ("stencil5", "stencil2d", (120, 100, 5, 500, 2)),
    ("kkt3d_12", "kkt3d", (12,)),
    ("banded_16k", "banded", (1 << 14, 32, 1024)),
]


@pytest.mark.parametrize("name,kind,args", DIRECT_CASES, ids=[c[0] for c in DIRECT_CASES])
def test_direct_shape(E, O, gpu, name, kind, args):
    """Small matrices with default sizing: one launch of the row-segment kernel, y assigned (whatever it
    held before), empty rows zeroed, long rows unsplit; also through the drop-in symbol."""
    cfg = E.make_config()
    c = Case(E, O, kind, args, cfg)
    plan = E.Plan(c.m, cfg)
    assert plan.stats["nnz_ell"] == 0 and plan.stats["nnz_er"] == c.nnz
    dx, dy = E.DeviceBuffer(c.n).upload(c.xp), E.DeviceBuffer(c.n).upload(np.full(c.n, 1e300))
    plan.spmv(dx.ptr, dy.ptr)
    y = dy.download()
    bad, worst = c.check(y)
    assert bad == 0, f"{name}: worst {worst:.3e}"
    plan.spmv(dx.ptr, dy.ptr)
    assert np.array_equal(dy.download(), y), "assigning kernel without atomics: bit-reproducible"
    with pytest.raises(E.EhybError):
        plan.spmv(dx.ptr, dy.ptr, phase=1)
    y1, it = E.spmv_gpu_ehyb(c.m, c.xp, 3)
    assert it == 3 and c.check(y1)[0] == 0


@pytest.mark.parametrize("mixed", [True, False], ids=["some_windows_kept", "no_window_kept"])
def test_pass_2_assigns_rows_of_partitions_without_a_window(E, O, gpu, mixed):
    """Pruned windows (plan.cpp): the ELL launch has no segment for those partitions and pass 2 of the panel residual
    is the only writer of their rows -- y is filled with NaN before every multiply, so a row nobody writes shows."""
    cfg = E.make_config(partitioner=E.EHYB_PART_CONTIGUOUS, er_mode=2, lds_doubles=4096)
    c = Case(E, O, None, None, cfg, matrix=fem_plus_rmat(E, cfg)) if mixed else Case(E, O, "rmat", (18, 1 << 21, 1), cfg)
    plan = E.Plan(c.m, cfg)
    u2 = plan.array("pb_units2").reshape(-1, 4)
    assert np.any(u2[:, 3] < 0) and (plan.stats["nnz_ell"] > 0) == mixed
    dx, dy = E.DeviceBuffer(c.n).upload(c.xp), E.DeviceBuffer(c.n)
    for phases in ((0,), (1, 2), (0,)):
        dy.upload(np.full(c.n, np.nan))
        for ph in phases:
            plan.spmv(dx.ptr, dy.ptr, phase=ph)
        y = dy.download()
        assert np.isfinite(y).all(), f"{int(np.isnan(y).sum())} rows were never written (phases {phases})"
        bad, worst = c.check(y)
        assert bad == 0, f"phases {phases}: worst {worst:.3e}"
