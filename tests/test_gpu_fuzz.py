"""The fuzz cases of test_fuzz_layout.py through the HIP kernels: every workgroup size, both
residual forms, symmetric pair storage on matrices that are and are not symmetric, windows from
64 doubles to 160 KiB, matrices from 1 x 1 up."""
import numpy as np
import pytest

from fuzz_cases import build

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", range(100, 140))
def test_random_plan_on_gpu(E, O, gpu, seed):
    m, cfg, kw, x, y_ref, scale = build(E, O, seed)
    plan = E.Plan(m, cfg)
    xp = E.vector_reorder(x, m.reorder_list)
    y = E.vector_recover(plan.spmv_host(xp, iters=2), m.reorder_list)
    bad, worst = O.check_tolerance(y, y_ref, scale)
    assert bad == 0, (kw, worst)
    # the two-phase call composes to the same result (a plan in the direct shape has no phases)
    dx, dy = E.DeviceBuffer(m.n).upload(xp), E.DeviceBuffer(m.n)
    if plan.stats["nnz_ell"] == 0 and plan.stats["nnz_er"] == plan.stats["nnz"] and plan.stats["er_segments"] == m.n:
        with pytest.raises(E.EhybError):
            plan.spmv(dx.ptr, dy.ptr, phase=1)
        return
    plan.spmv(dx.ptr, dy.ptr, phase=1)
    plan.spmv(dx.ptr, dy.ptr, phase=2)
    assert O.check_tolerance(E.vector_recover(dy.download(), m.reorder_list), y_ref, scale)[0] == 0, kw
