"""ehyb_plan_tune: the item -> workgroup map of a plan re-made from per-XCD rates measured on the device (the heaviest work
items on the XCDs that streamed fastest).  Whatever map it ends with, every item is taken exactly once: the product is right."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kw,gen", [
    (dict(sym_pairs=1), ("fem3d", 196608, 3, 42, 42, 13500, 1, 1)),          # one workgroup per partition, one round
    (dict(sym_pairs=0, direct=2), ("fem3d", 196608, 3, 42, 42, 13500, 1, 1)),  # plain storage: default map = one run of items per XCD
    (dict(sym_pairs=1, lds_doubles=2048), ("fem3d", 120000, 3, 35, 35, 13500, 1, 1)),   # several rounds of workgroups: a no-op
], ids=["sym-one-round", "plain-one-round", "sym-several-rounds"])
def test_tuned_plan_multiplies_right(E, O, gpu, kw, gen):
    cfg = E.make_config(**kw)
    m = E.Matrix.generate(*gen, cfg=cfg)
    n = m.n
    x = O.x_glibc(n)
    y_ref = O.spmv_coo(n, m.I, m.J, m.V, x)
    scale = O.abs_rowsum(n, m.I, m.J, m.V, x)
    m.reorder(cfg)
    perm = m.reorder_list.copy()
    plan = E.Plan(m, cfg)
    dx, dy = E.DeviceBuffer(n).upload(E.vector_reorder(x, perm)), E.DeviceBuffer(n)
    st = plan.stats
    for rnd in range(2):                       # a second call starts from the first one's map
        before, after = plan.tune(dx.ptr, dy.ptr, reps=3)
        if st["n_items"] > 256 * int(cfg.items_per_cu):
            assert before == 0 and after == 0  # more than one round of workgroups: left alone
        else:
            assert before > 0 and 0 < after <= before
        dy.upload(np.full(n, np.nan))
        plan.spmv(dx.ptr, dy.ptr)
        y = E.vector_recover(dy.download(), perm)
        bad, worst = O.check_tolerance(y, y_ref, scale)
        assert bad == 0, (rnd, worst)
