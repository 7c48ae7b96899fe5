"""Committed fixtures (tests/golden): reader + oracle + layout walk, CPU only; the same
fixtures through the HIP path under -m gpu."""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden")
NAMES = ["sym_grid", "gen_band", "pat_graph"]


def _load(E, O, name, cfg=None):
    m = E.Matrix.read_mtx(os.path.join(GOLD, name + ".mtx"), cfg)
    y = np.load(os.path.join(GOLD, name + ".y.npy"))
    scale = np.load(os.path.join(GOLD, name + ".scale.npy"))
    return m, y, scale


@pytest.mark.parametrize("name", NAMES)
def test_oracle_reproduces_fixture(E, O, name):
    m, y_gold, scale = _load(E, O, name)
    x = O.x_glibc(m.n)
    y = O.spmv_coo(m.n, m.I, m.J, m.V, x)
    assert O.check_tolerance(y, y_gold, scale)[0] == 0
    assert np.allclose(O.abs_rowsum(m.n, m.I, m.J, m.V, x), scale, rtol=1e-13, atol=1e-300)
    assert m.symmetric == (name == "sym_grid")


@pytest.mark.parametrize("name", NAMES)
@pytest.mark.parametrize("mode", [1, 2])
def test_layout_walk_reproduces_fixture(E, O, name, mode):
    cfg = E.make_config(window_mode=mode, lds_doubles=128)
    m, y_gold, scale = _load(E, O, name, cfg)
    x = O.x_glibc(m.n)
    m.reorder(cfg)
    plan = E.Plan(m, cfg, upload=False)
    yp, written = O.walk_plan(plan, E.vector_reorder(x, m.reorder_list))
    assert (written == 1).all()
    y = E.vector_recover(yp, m.reorder_list)
    assert O.check_tolerance(y, y_gold, scale)[0] == 0


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_gpu_reproduces_fixture(E, O, gpu, name):
    cfg = E.make_config(lds_doubles=128)
    m, y_gold, scale = _load(E, O, name, cfg)
    x = O.x_glibc(m.n)
    m.reorder(cfg)
    yp, _ = E.spmv_gpu_ehyb(m, E.vector_reorder(x, m.reorder_list), 2)
    y = E.vector_recover(yp, m.reorder_list)
    bad, worst = O.check_tolerance(y, y_gold, scale)
    assert bad == 0, f"{name}: worst {worst:.3e}"
