"""Fuzz of the host half: random small matrices (empty, one row, dense rows/columns, symmetric,
block-structured) x random plan configurations (window mode and size, workgroup size, residual
form, symmetric pair storage, column sharing, item count); every layout is walked the way the
kernels index it and compared with the oracle product.  700 further seeds were run when this
test was written; the committed ones keep the CPU suite short."""
import pytest

from fuzz_cases import build


@pytest.mark.parametrize("seed", range(48))
def test_random_layout_walk(E, O, seed):
    m, cfg, kw, x, y_ref, scale = build(E, O, seed)
    plan = E.Plan(m, cfg, upload=False)
    yp, written = O.walk_plan(plan, E.vector_reorder(x, m.reorder_list))
    assert (written == 1).all(), kw
    bad, worst = O.check_tolerance(E.vector_recover(yp, m.reorder_list), y_ref, scale)
    assert bad == 0, (kw, worst)
    st = plan.stats
    assert st["nnz_ell"] + st["nnz_er"] == m.nnz, kw


@pytest.mark.parametrize("seed", range(200, 224))
def test_random_plan_cache_round_trip(E, O, seed, tmp_path):
    """ehyb_plan_save / ehyb_plan_load on the same random plans: every array comes back bit for bit,
    and so does the permutation."""
    import numpy as np

    m, cfg, kw, x, y_ref, scale = build(E, O, seed)
    plan = E.Plan(m, cfg, upload=False)
    path = tmp_path / "p.cache"
    plan.save(path, reorder_list=m.reorder_list, key=12345)
    back, perm = E.Plan.load(path, key=12345, upload=False)
    assert np.array_equal(perm, m.reorder_list), kw
    for name in E.ARRAYS:
        if name != "perm":
            assert np.array_equal(plan.array(name), back.array(name)), (name, kw)
    assert back.stats == plan.stats, kw
    yp, written = O.walk_plan(back, E.vector_reorder(x, perm))
    assert (written == 1).all() and O.check_tolerance(E.vector_recover(yp, perm), y_ref, scale)[0] == 0, kw
