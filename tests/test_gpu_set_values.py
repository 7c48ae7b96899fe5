"""ehyb_plan_set_values on the GPU (csrc/ehyb_fill.hip): the numeric phase of the build -- the
value fill of convert.c:316-369 and, through entry_order, the V scatter of reordering.c:348-362 -- repeated on
the device for new values on the pattern a plan was built from.  The refilled plan must multiply like a plan
built from scratch from the new matrix: against the CPU oracle on the new values, and bit for bit against the
fresh plan where the multiply is deterministic (plain storage)."""
import ctypes as C

import numpy as np
import pytest

from test_value_map import CASES
from util import Case

pytestmark = pytest.mark.gpu


def new_values(I, J, symmetric):
    """Deterministic new values on the same pattern; symmetric = a function of the unordered pair (i, j)."""
    a, b = (np.minimum(I, J), np.maximum(I, J)) if symmetric else (I, J)
    h = (a.astype(np.int64) * 2654435761 + b.astype(np.int64) * 40503 + 12345) % 2003
    return (h - 1001).astype(np.float64) / 977.0 + 0.0005


@pytest.mark.parametrize("name,kind,args,kw", CASES, ids=[c[0] for c in CASES])
@pytest.mark.parametrize("how", ["permuted_host", "original_host", "original_device"])
def test_refilled_plan_multiplies_like_a_fresh_one(E, O, gpu, name, kind, args, kw, how):
    cfg = E.make_config(value_map=1, **kw)
    m0 = E.Matrix.generate(kind, *args, cfg=cfg)
    I0, J0, rp0 = m0.I.copy(), m0.J.copy(), m0.row_idx.copy()
    sym = kw.get("sym_pairs", 0) == 1
    V2_orig = new_values(I0, J0, sym)
    m0.free()
    c = Case(E, O, kind, args, cfg)                 # the plan is built from the generator's own values
    plan = E.Plan(c.m, cfg)
    bad, worst = c.check(plan.spmv_host(c.xp))
    assert bad == 0, f"before the refill: worst {worst:.3e}"
    order = E.entry_order(rp0, c.perm)
    # the oracle on the NEW matrix, original numbering (solver_test.c:102)
    y2_ref = O.spmv_coo(c.n, I0, J0, V2_orig, c.x)
    scale2 = O.abs_rowsum(c.n, I0, J0, V2_orig, c.x)
    if how == "permuted_host":
        plan.set_values(V2_orig[order])
    elif how == "original_host":
        plan.set_values(V2_orig, entry_order=order)
    else:
        lib = plan.lib
        dv, do = C.c_void_p(), C.c_void_p()
        assert lib.ehyb_dev_alloc(V2_orig.nbytes, C.byref(dv)) == 0 and lib.ehyb_dev_alloc(order.nbytes, C.byref(do)) == 0
        assert lib.ehyb_h2d(dv, V2_orig.ctypes.data_as(C.c_void_p), V2_orig.nbytes) == 0
        assert lib.ehyb_h2d(do, order.ctypes.data_as(C.c_void_p), order.nbytes) == 0
        plan.set_values((dv.value, len(V2_orig)), entry_order=(do.value, len(order)))
        assert lib.ehyb_dev_sync() == 0
        lib.ehyb_dev_free(dv), lib.ehyb_dev_free(do)
    y2p = plan.spmv_host(c.xp)
    bad, worst = O.check_tolerance(c.recover(y2p), y2_ref, scale2)
    assert bad == 0, f"{name}/{how}: {bad} rows over tolerance after the refill, worst {worst:.3e}"
    # a plan built from scratch from the new matrix: same pattern, same layout
    c.m.V[:] = V2_orig[order]
    fresh = E.Plan(c.m, cfg)
    assert fresh.stats == plan.stats
    yf = fresh.spmv_host(c.xp)
    if not sym and plan.stats["er_partials"] == 0:   # (LDS atomics -- symmetric pairs, panel residual -- add in any order)
        assert np.array_equal(yf.view(np.int64), y2p.view(np.int64)), "refilled and fresh plan differ bitwise"
    else:
        assert np.all(np.abs(yf - y2p) <= 1e-12 * np.maximum(E.vector_reorder(scale2, c.perm), 1e-300))
    # phases too (residual arrays of an inline plan are refilled as well)
    if not plan.stats["er_inline"] and kw.get("direct", 0) != 1 and plan.stats["nnz_er"] > 0:
        dx, dy = E.DeviceBuffer(c.n).upload(c.xp), E.DeviceBuffer(c.n)
        plan.spmv(dx.ptr, dy.ptr, phase=1)
        plan.spmv(dx.ptr, dy.ptr, phase=2)
        assert plan.lib.ehyb_dev_sync() == 0
        bad, worst = O.check_tolerance(c.recover(dy.download()), y2_ref, scale2)
        assert bad == 0
    plan.destroy(), fresh.destroy()


def test_unsymmetric_values_on_a_symmetric_plan_are_refused(E, O, gpu):
    cfg = E.make_config(value_map=1, lds_doubles=4096, sym_pairs=1)
    c = Case(E, O, "fem3d", (30000, 3, 22, 22, 13500, 1, 1), cfg)
    plan = E.Plan(c.m, cfg)
    assert plan.stats["sym_pairs"] > 0
    y_before = plan.spmv_host(c.xp)
    with pytest.raises(E.EhybError) as ei:
        plan.set_values(new_values(c.m.I, c.m.J, symmetric=False))
    assert ei.value.code == 1 and "a_ij != a_ji" in str(ei.value)
    assert np.all(np.abs(plan.spmv_host(c.xp) - y_before) <= 1e-12 * np.maximum(np.abs(y_before), 1.0))   # plan unchanged
    plan.set_values(new_values(c.m.I, c.m.J, symmetric=True))                                     # symmetric ones are taken
    plan.destroy()


def test_bad_arguments(E, O, gpu, tmp_path):
    cfg = E.make_config(value_map=1, lds_doubles=4096, direct=2)
    c = Case(E, O, "fem3d", (30000, 3, 22, 22, 13500, 1, 1), cfg)
    plan = E.Plan(c.m, cfg)
    V = c.m.V.copy()
    with pytest.raises(E.EhybError):            # wrong length
        plan.set_values(V[:-1])
    bad_order = np.arange(len(V), dtype=np.int32)
    bad_order[7] = len(V)
    with pytest.raises(E.EhybError) as ei:      # entry_order out of range: caught on the device before any write
        plan.set_values(V, entry_order=bad_order)
    assert "outside" in str(ei.value)
    plan.save(tmp_path / "ok.plan")             # host copy still current
    plan.set_values(2.0 * V)
    with pytest.raises(E.EhybError) as ei:      # host copy stale
        plan.save(tmp_path / "stale.plan")
    assert ei.value.code == 8
    bad, worst = O.check_tolerance(c.recover(plan.spmv_host(c.xp)), 2.0 * c.y_ref, 2.0 * c.scale)
    assert bad == 0
    # a plan without maps, and one that is not on the device
    plain = E.Plan(c.m, E.make_config(lds_doubles=4096, direct=2))
    with pytest.raises(E.EhybError) as ei:
        plain.set_values(V)
    assert ei.value.code == 8
    host_only = E.Plan(c.m, cfg, upload=False)
    with pytest.raises(E.EhybError) as ei:
        host_only.set_values(V)
    assert ei.value.code == 8
