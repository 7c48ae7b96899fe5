"""The drop-in boundary on the GPU:
  * the reference's own driver binary (solver_test.c unchanged + include/ + libehyb.so, built by
    oracle/Makefile where the reference tree exists) reads a .mtx, forms its own CPU product and judges
    this library's GPU result with its own compare();
  * the reference's sizing hints (audikw_1: nParts = 164, cache 6144) through the reference-named
    calls matrixReorder -> vectorReorder -> spmvGPuEHYB -> vectorRecover give the same storage and
    the same speed as the plan API that bench.py times."""
import os
import re
import subprocess

import numpy as np
import pytest

from util import Case

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER = os.path.join(ROOT, "oracle", "_ref", "solver_test_ref")

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kind", ["symmetric", "general"])
def test_reference_driver_end_to_end(E, gpu, tmp_path, kind):
    if not os.path.exists(DRIVER):
        pytest.skip("oracle/_ref/solver_test_ref was not built (needs the reference tree at build time)")
    (tmp_path / "read").mkdir()
    if kind == "symmetric":
        m = E.Matrix.generate("fem3d", 120000, 3, 35, 35, 13500, 1, 1)
        m.write_mtx(tmp_path / "read" / "a.mtx", symmetric_lower_only=True)
    else:
        m = E.Matrix.generate("rmat", 16, 1 << 19, 3)       # the driver prints y[30000..30009]: needs > 30,010 rows
        m.write_mtx(tmp_path / "read" / "a.mtx")
    p = subprocess.run([DRIVER, "-m", "a", "-i", "50"], cwd=tmp_path, capture_output=True, text=True, timeout=600)
    out = p.stdout
    assert p.returncode == 0, out[-1500:] + p.stderr[-1500:]
    assert ("read symmetric matrix" if kind == "symmetric" else "unsymmetric reordering") in out
    assert "sizeER is" in out and "iter is 50, time is" in out                      # spmv.cu:82,121
    # solver_test.c:15-21 flags rows that differ by more than 1 % of min(|y|, |yResult|): with a different
    # summation order that trips on rows whose terms cancel to rounding level (both print as +-0.000000)
    for a, b in re.findall(r"large difference at \d+\s*: realy (-?[0-9.]+) vs yResult (-?[0-9.]+)", out):
        assert abs(float(a)) < 1e-6 and abs(float(b)) < 1e-6, (a, b)
    diff = float(re.search(r"diff is ([0-9.eE+-]+)", out).group(1))                   # solver_test.c:28: sum |dy|
    total = float(np.abs(m.V).sum()) * 0.1
    assert diff <= 1e-12 * total, (diff, total)
    # the ten rows it prints agree to the printed digits
    for a, b in re.findall(r"yResult is (-?[0-9.]+) y is  (-?[0-9.]+)", out):
        assert a == b


def test_reference_sizing_through_the_reference_calls(E, O, gpu):
    """audikw_1-like matrix with the reference driver's own sizing (solver_test.c:158-182: 164
    partitions, cache 6144) through matrixReorder / spmvGPuEHYB as the driver calls them: right result,
    symmetric pair storage chosen without any configuration, and the speed of the plan path."""
    cfg = E.make_config(sym_pairs=1)
    c = Case(E, O, "fem3d", (943695, 3, 68, 68, 13500, 1, 1), cfg, reorder=False)
    m = c.m
    m.c.nParts, m.c.vectorCacheSize, m.c.kernelPerPart = 164, 6144, 0
    m.reorder_dropin()                                                  # matrixReorder(&m)
    c.perm = m.reorder_list.copy()
    c.xp = E.vector_reorder(c.x, c.perm)
    assert m.c.nParts % 256 == 0 or m.c.nParts > 256
    y, it, ms = E.spmv_gpu_ehyb(m, c.xp, 200, timing=True)              # spmvGPuEHYB(&m, x, y, 200, &it)
    assert it == 200 and c.check(y)[0] == 0
    gflops_dropin = 2.0 * c.nnz * it / (ms * 1e6)
    plan = E.Plan(m, cfg)
    dx, dy = E.DeviceBuffer(c.n).upload(c.xp), E.DeviceBuffer(c.n)
    r = plan.bench(dx.ptr, dy.ptr, warmup=10, iters=200, per_kernel=False)
    gflops_plan = 2.0 * c.nnz * 200 / (r["ms_total"] * 1e6)
    assert plan.stats["sym_pairs"] > 0
    print(f"drop-in {gflops_dropin:.1f} GFLOP/s, plan API {gflops_plan:.1f} GFLOP/s")
    assert gflops_dropin > 0.9 * gflops_plan
