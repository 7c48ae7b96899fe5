"""The unstructured-mesh generator (ehyb_gen_mesh3d: random points, k nearest neighbours, d unknowns per node) -- the
surrogate that is NOT a lattice -- and `bench.py --mtx` on a file written from it."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_mesh3d_is_symmetric_unstructured_and_walks_to_the_reference_product(E, O):
    n, dof, knn = 30000, 3, 12
    m = E.Matrix.generate("mesh3d", n, dof, knn, 1500, 7)
    A = m.to_scipy()
    assert m.n == n and abs(A - A.T).max() == 0 and (A.diagonal() != 0).all()
    rl = np.diff(A.indptr)
    assert rl.min() >= (knn + 1) * dof and rl.max() > rl.min() + 3 * dof and rl.max() <= (3 * knn + 1) * dof   # degrees vary, no fixed stencil
    # the unknowns of a node share their column list (what the layout shares column words for)
    rp, J = A.indptr, A.indices
    assert all(np.array_equal(J[rp[r]:rp[r + 1]], J[rp[r + 1]:rp[r + 2]]) for r in range(0, 300, 3))
    # labels carry no locality: the neighbours of a row are spread over the whole index range
    spread = np.abs(J[rp[0]:rp[1]].astype(np.int64) - 0)
    assert spread.max() > n // 4
    m2 = E.Matrix.generate("mesh3d", n, dof, knn, 1500, 7)
    assert np.array_equal(m2.J, m.J) and np.array_equal(m2.V, m.V)        # deterministic
    for kw in (dict(sym_pairs=1, lds_doubles=4096), dict(lds_doubles=2048)):
        cfg = E.make_config(**kw)
        g = E.Matrix.generate("mesh3d", n, dof, knn, 1500, 7, cfg=cfg)
        x = O.x_glibc(n)
        y_ref = O.spmv_coo(n, g.I, g.J, g.V, x)
        scale = O.abs_rowsum(n, g.I, g.J, g.V, x)
        g.reorder(cfg)
        plan = E.Plan(g, cfg, upload=False)
        yp, written = O.walk_plan(plan, E.vector_reorder(x, g.reorder_list))
        assert (written == 1).all()
        assert O.check_tolerance(E.vector_recover(yp, g.reorder_list), y_ref, scale)[0] == 0
        if kw.get("sym_pairs"):
            assert plan.stats["sym_pairs"] > 0.25 * plan.stats["nnz"] / 2


@pytest.mark.gpu
def test_bench_on_a_matrix_market_file(E, gpu, tmp_path):
    """bench.py --mtx PATH: the headline line on a real file (here: one written from the mesh generator, symmetric,
    lower triangle) -- storage from the banner, parity against the oracle, `data` says which file."""
    m = E.Matrix.generate("mesh3d", 60000, 3, 12, 1500, 3)
    path = tmp_path / "mesh60k.mtx"
    m.write_mtx(path, symmetric_lower_only=True)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--mtx", str(path), "--steps", "20", "--warmup", "3", "--no-scaling-anchor"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-2500:]
    out = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][0])
    assert out["data"].startswith("file: ") and "mesh60k.mtx" in out["data"] and out["config"]["workload"] == "mesh60k"
    assert out["config"]["rows"] == 60000 and out["config"]["nnz"] == m.nnz and out["config"]["sym_pairs"] > 0
    assert out["parity"]["rows_over_1e-12"] == 0 and out["value"] > 0
    assert "plain_storage" not in out and "dropin_path" not in out
