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


def _ref_cases():
    import importlib.util

    spec = importlib.util.spec_from_file_location("ref_cases", os.path.join(GOLD, "ref_cases.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


# the reference driver's own sizing for each case (solver_test.c:78,183: its 82-SM heuristic ran, not this library's)
REF_DRIVER_SIZING = {"sym": "parts is 16 with cachSize 10240", "general": "parts is 10 with cachSize 8192"}
REF_DRIVER_TAGS = [t for t in ("sym", "general", "general_zeros", "sym_zeros") if os.path.exists(os.path.join(GOLD, f"ref_driver_{t}.txt"))]


@pytest.mark.parametrize("tag", REF_DRIVER_TAGS)
def test_oracle_against_the_reference_drivers_own_output(E, O, tag, tmp_path):
    """tests/golden/ref_driver_<tag>.txt holds what the REFERENCE'S OWN driver printed (solver_test.c
    unchanged, on top of libehyb.so, run on a GPU box by make_ref_driver_golden.py): its sizing line
    (solver_test.c:78,183), rows 30000-30009 of its CPU product `y` (solver_test.c:102 / 247,254 -- the
    code oracle/ehyb_oracle.c restates) next to this library's GPU result, up to 100 more rows its own 1 % test
    flags (cancellation rows: "large difference at i : realy <y> vs yResult <GPU>"), and its compare() line.
    The oracle reproduces the reference's y to the printed digits on every printed row; the reference judged the GPU
    result equal to its own (identical columns, summed difference at rounding level).  Cases: tests/golden/ref_cases.py
    (symmetric and general readers, explicit zeros, zero diagonal entries, mixed magnitudes)."""
    import re

    text = open(os.path.join(GOLD, f"ref_driver_{tag}.txt")).read()
    if tag in REF_DRIVER_SIZING:
        assert REF_DRIVER_SIZING[tag] in text
    assert re.search(r"parts is \d+ with cachSize \d+", text)
    rows = re.findall(r"at (\d+) yResult is (-?[0-9.]+) y is\s+(-?[0-9.]+)", text)
    assert len(rows) >= 9
    flagged = re.findall(r"large difference at (\d+)\s*: realy (-?[0-9.]+) vs yResult (-?[0-9.]+)", text)
    # the matrix as the driver read it: through the same Matrix Market file
    m0, sym = _ref_cases().build(E, tag)
    path = tmp_path / "a.mtx"
    m0.write_mtx(path, symmetric_lower_only=sym)
    m = E.Matrix.read_mtx(path)
    assert m.symmetric == sym
    x = O.x_glibc(m.n)
    y = O.spmv_coo(m.n, m.I, m.J, m.V, x)

    def same(a, b):
        return a.replace("-0.000000", "0.000000") == b.replace("-0.000000", "0.000000")

    for i, y_gpu, y_ref in rows:
        assert same(f"{y[int(i)]:.6f}", y_ref), (i, y[int(i)], y_ref)
        assert same(y_gpu, y_ref)
    for i, y_ref, y_gpu in flagged:          # rows whose terms cancel: |y| tiny against its terms, both print alike
        assert same(f"{y[int(i)]:.6f}", y_ref), (i, y[int(i)], y_ref)
        assert same(y_gpu, y_ref)
    diff = float(re.search(r"diff is ([0-9.eE+-]+)", text).group(1))    # sum over ALL rows of |y - yResult|
    scale = float(np.abs(O.abs_rowsum(m.n, m.I, m.J, m.V, x)).sum())
    assert diff < 1e-11 or diff < 1e-13 * scale
