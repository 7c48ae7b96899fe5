"""The harness binary end to end (reference usage: ./spmv.out -i 2000 -m audikw_1, README.md:10):
`solver_test -m <name> -i <iters>` reads ./read/<name>.mtx, follows the banner for symmetric vs
general, runs the whole path and compares with its in-line CPU product."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "ehyb_spmv_gpu_amd", "solver_test")

pytestmark = pytest.mark.gpu


def _run(args, cwd):
    return subprocess.run([BIN] + args, cwd=cwd, capture_output=True, text=True, timeout=600)


def test_cli_symmetric_mtx_bcsstk17_sized(E, gpu, tmp_path):
    """BASELINE config 1's size (bcsstk17: 10,974 rows, 428,650 entries) as a symmetric .mtx."""
    (tmp_path / "read").mkdir()
    m = E.Matrix.generate("fem3d", 10974, 3, 62, 59, 250000, 1, 17)   # one 62 x 59 layer of 3-dof nodes
    assert m.n == 10974 and 350_000 < m.nnz < 520_000
    m.write_mtx(tmp_path / "read" / "bcsstk17_like.mtx", symmetric_lower_only=True)
    p = _run(["-i", "200", "-m", "bcsstk17_like"], tmp_path)
    out = p.stdout
    assert p.returncode == 0, out[-2000:] + p.stderr[-2000:]
    assert "filename is ./read/bcsstk17_like.mtx" in out and "read symmetric matrix" in out
    assert "sizeER is" in out and "iter is 200, time is" in out and "GPU Gflops is" in out   # spmv.cu:82,121
    assert "diff is" in out and "PASSED" in out                                             # solver_test.c:28


def test_cli_general_mtx_and_reference_window(E, gpu, tmp_path):
    (tmp_path / "read").mkdir()
    m = E.Matrix.generate("rmat", 12, 1 << 15, 3)
    m.write_mtx(tmp_path / "read" / "g.mtx")
    p = _run(["-i", "20", "-m", "g", "-w", "1", "-l", "1024"], tmp_path)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert "read unsymmetric matrix" in p.stdout and "PASSED" in p.stdout


def test_cli_usage_errors(gpu, tmp_path):
    p = _run(["-i", "10"], tmp_path)                       # no matrix: solver_test.c:318-321
    assert p.returncode != 0 and "file name or max iteration number missing" in p.stdout
    p = _run(["-i", "10", "-m", "does_not_exist"], tmp_path)
    assert p.returncode != 0 and "file read error" in p.stdout    # solver_test.c:328-331
    p = _run(["-i", "5", "-g", "stencil2d:40:30:5:100"], tmp_path)
    assert p.returncode == 0 and "PASSED" in p.stdout


def test_cli_plan_cache_miss_then_hit(E, gpu, tmp_path):
    """-c file: the first run partitions, converts and writes the cache; the second reads the
    permutation and the layout back and skips both steps; a different matrix is a miss."""
    cache = str(tmp_path / "plan.cache")
    a = ["-i", "20", "-g", "fem3d:30000:3:22:22:13500:1", "-c", cache]
    p1 = _run(a, tmp_path)
    assert p1.returncode == 0, p1.stdout[-2000:] + p1.stderr[-2000:]
    assert "plan cache miss" in p1.stdout and "reorder time is" in p1.stdout and "plan cache written" in p1.stdout
    assert "PASSED" in p1.stdout and "iter is 20, time is" in p1.stdout
    p2 = _run(a, tmp_path)
    assert p2.returncode == 0, p2.stdout[-2000:] + p2.stderr[-2000:]
    assert "plan cache hit" in p2.stdout and "reorder time is" not in p2.stdout and "PASSED" in p2.stdout
    p3 = _run(["-i", "5", "-g", "fem3d:30000:3:22:22:13400:1", "-c", cache], tmp_path)   # other matrix, same file
    assert p3.returncode == 0 and "plan cache miss" in p3.stdout and "another matrix" in p3.stdout and "PASSED" in p3.stdout


@pytest.mark.parametrize("gen,items", [("fem3d:99000:3:30:30:13500:1", 700), ("rmat:15:300000", 13)],
                         ids=["fem3d-many-items", "rmat-odd-items"])
def test_cli_item_order_switch(gpu, tmp_path, gen, items):
    """solver_test -X 0/1 (cfg.xcd_map): workgroups take the work items in blockIdx order or one contiguous run per
    XCD (the default).  Either way every item is taken exactly once -- including item counts that
    are not a multiple of 8 -- so both runs pass the harness's comparison with the CPU product."""
    for xcd in ("0", "1"):
        p = subprocess.run([BIN, "-i", "5", "-g", gen, "-l", "2048", "-X", xcd, "-I", str(max(1, items // 256))], cwd=tmp_path,
                           capture_output=True, text=True, timeout=600)
        assert p.returncode == 0 and "PASSED" in p.stdout, (xcd, p.stdout[-1500:] + p.stderr[-1500:])


def test_bench_vendor_baseline_arm(gpu):
    """bench.py --vendor-baseline: rocSPARSE CSR SpMV (the role of the reference's cuSPARSE baselines,
    spmv.cu:135-281) on the same matrix and GPU, every algorithm checked against the CPU oracle, reported
    beside the EHYB number in the same JSON line.  Opt-in: the product path never links rocSPARSE."""
    import json
    import sys

    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "small", "--steps", "20", "--warmup", "3",
                        "--vendor-baseline", "--no-cpu-baseline", "--no-plain-arm", "--no-live-pmc"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-2500:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    v = out["vendor_baseline"]
    assert v["best"]["rows_over_1e-12"] == 0 and v["best"]["GFLOP/s"] > 0
    assert out["parity"]["rows_over_1e-12"] == 0 and out["scaling"] == "none"
    ldd = subprocess.run(["ldd", os.path.join(ROOT, "ehyb_spmv_gpu_amd", "libehyb.so")], capture_output=True, text=True).stdout
    assert "rocsparse" not in ldd


def test_bench_measures_its_traffic_in_the_run(gpu):
    """roofline.traffic is MEASURED by the run that prints it: bench.py starts tools/pmc_run.py twice under
    `rocprofv3 --pmc` (FETCH_SIZE, WRITE_SIZE; separate passes, with --kernel-trace), after the timed loop, and quotes the
    bytes only for a child plan with its own layout fingerprint."""
    import json
    import sys

    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "small", "--steps", "20", "--warmup", "3",
                        "--no-cpu-baseline", "--no-plain-arm", "--no-dropin-arm", "--no-scaling-anchor", "--no-refill-arm"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-2500:]
    out = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][0])
    r = out["roofline"]
    assert r["bytes_basis"].startswith("rocprofv3 PMC bytes per launch measured by this run"), (r["bytes_basis"], p.stderr[-1500:])
    assert r["traffic_live"]["launches"] >= 5 and 1.9 < r["traffic_live"]["fetch_factor"] < 2.1
    # what the counters see is what the format says the kernel moves, to the partial cache lines at the ends of its streams
    assert 0.9 * r["format_bytes_per_launch"] < r["traffic"] < 1.25 * r["format_bytes_per_launch"], (r["traffic"], r["format_bytes_per_launch"])
    assert out["parity"]["rows_over_1e-12"] == 0
