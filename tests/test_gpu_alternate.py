"""Successive multiplies of a plan walk their streams in alternating directions (cfg.ell_alternate; ELL slabs and pass 1 of the
panel residual): the direction changes the order of the work, never the result."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _case(E, O, kind, args, cfg):
    m = E.Matrix.generate(kind, *args, cfg=cfg)
    x = O.x_glibc(m.n)
    y_ref = O.spmv_coo(m.n, m.I, m.J, m.V, x)
    scale = O.abs_rowsum(m.n, m.I, m.J, m.V, x)
    m.reorder(cfg)
    return m, x, y_ref, scale


@pytest.mark.parametrize("name,kind,args,kw,deterministic", [
    ("fem-plain", "fem3d", (60000, 3, 28, 28, 13500, 1, 1), dict(lds_doubles=4096), True),
    ("fem-reference-window", "fem3d", (60000, 3, 28, 28, 13500, 1, 1), dict(lds_doubles=1024, window_mode=1), False),
    ("fem-symmetric-pairs", "fem3d", (60000, 3, 28, 28, 13500, 1, 1), dict(lds_doubles=4096, sym_pairs=1), False),
    ("banded-relative-columns", "banded", (1024 * 64, 32, 1024), dict(), True),
    ("rmat-panel", "rmat", (17, 1 << 20, 3), dict(er_mode=2, fuse_er=2, direct=2, er_panel_cols=4096), False),
    ("rmat-panel-units", "rmat", (16, 1 << 19, 5), dict(er_mode=2, fuse_er=2, direct=2, er_units1=7, er_panel_cols=1024), False),
], ids=lambda v: v if isinstance(v, str) else None)
def test_both_directions_give_the_product(E, O, gpu, name, kind, args, kw, deterministic):
    cfg = E.make_config(ell_alternate=1, **kw)          # (1 = always: these matrices are far smaller than the cache)
    m, x, y_ref, scale = _case(E, O, kind, args, cfg)
    perm = m.reorder_list
    plan = E.Plan(m, cfg)
    dx = E.DeviceBuffer(m.n).upload(E.vector_reorder(x, perm))
    ys = []
    for _ in range(4):                                   # first to last, last to first, and again
        dy = E.DeviceBuffer(m.n)
        dy.upload(np.full(m.n, np.nan))
        plan.spmv(dx.ptr, dy.ptr)
        E.host._lib.load().ehyb_dev_sync()
        ys.append(dy.download())
        bad, worst = O.check_tolerance(E.vector_recover(ys[-1], perm), y_ref, scale)
        assert bad == 0, (name, len(ys), worst)
    if deterministic:                                    # a row's sum does not depend on the order its slab is taken in
        assert np.array_equal(ys[0], ys[1]) and np.array_equal(ys[0], ys[2])
    never = E.Plan(m, E.make_config(ell_alternate=2, **kw))
    dy = E.DeviceBuffer(m.n)
    never.spmv(dx.ptr, dy.ptr)
    E.host._lib.load().ehyb_dev_sync()
    if deterministic:
        assert np.array_equal(dy.download(), ys[1])
