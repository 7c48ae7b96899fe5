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
