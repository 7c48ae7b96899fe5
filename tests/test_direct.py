"""Direct shape for small matrices (cfg.direct; the role of the reference's small-matrix branch,
kernel.cu:197-284 / solver_test.c:56-69): no LDS window, every row one segment of the row-segment
kernel, which ASSIGNS y -- one small launch.  Host side checked by the oracle's walk; the GPU side is
tests/test_gpu_parity.py::test_direct_shape."""
import numpy as np
import pytest

from util import Case

CASES = [
    ("bcsstk17_like", "fem3d", (10974, 3, 62, 59, 250000, 1, 17)),
    ("rmat_s13_hubs", "rmat", (13, 1 << 18, 3)),          # empty rows and rows of > 4096 entries
    ("stencil5", "stencil2d", (120, 100, 5, 500, 2)),
    ("kkt3d_12", "kkt3d", (12,)),
]


@pytest.mark.parametrize("name,kind,args", CASES, ids=[c[0] for c in CASES])
def test_direct_layout_walks_to_the_reference_product(E, O, name, kind, args):
    cfg = E.make_config()                                  # defaults: automatic for <= EHYB_DIRECT_MAX_ROWS rows
    c = Case(E, O, kind, args, cfg)
    assert c.n <= 81920
    plan = E.Plan(c.m, cfg, upload=False)
    st = plan.stats
    assert st["nnz_ell"] == 0 and st["nnz_er"] == c.nnz and st["size_block_ell"] == 0 and st["er_partials"] == 0
    seg_row = plan.array("er_seg_row")
    assert len(seg_row) == c.n and np.array_equal(np.sort(seg_row), np.arange(c.n)), "one unsplit segment per row, empty rows included"
    y, written = O.walk_plan(plan, c.xp)
    bad, worst = c.check(y)
    assert bad == 0, f"worst {worst:.3e}"


def test_direct_is_only_automatic_for_default_windows(E, O):
    small = ("stencil2d", (120, 100, 5, 500, 2))
    for kw, want in ((dict(), True), (dict(direct=2), False), (dict(lds_doubles=4096), False), (dict(sym_pairs=1), False),
                     (dict(window_mode=1), False), (dict(lds_doubles=4096, direct=1), True)):
        cfg = E.make_config(**kw)
        c = Case(E, O, *small, cfg)
        st = E.Plan(c.m, cfg, upload=False).stats
        assert (st["nnz_ell"] == 0) == want, kw
    # larger matrices keep the window
    cfg = E.make_config()
    c = Case(E, O, "fem3d", (90000, 3, 32, 32, 13500, 1, 1), cfg)
    assert E.Plan(c.m, cfg, upload=False).stats["nnz_ell"] > 0
