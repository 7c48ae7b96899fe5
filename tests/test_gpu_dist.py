"""The N > 1 paths on one GPU: `bench.py --gpus 2` with both ranks on cuda:0 over gloo (a functional
run -- RCCL refuses ranks that share a device).  The HIP kernels, the two-phase multiply, the
stream overlap and the exchange logic are the ones the multi-GPU run uses; only the transport
differs.  bench.py checks its own result against the CPU oracle and refuses to print otherwise."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _bench(world, extra):
    env = dict(os.environ, EHYB_BENCH_ONE_DEVICE="1", EHYB_BENCH_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"),
           "--gpus", str(world), "--steps", "10", "--warmup", "2", "--workload", "small"] + extra
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    return json.loads(p.stdout.strip().splitlines()[-1])


def test_weak_halo_exchange_two_ranks(gpu):
    out = _bench(2, [])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak"
    assert out["parity"]["rows_over_1e-12"] == 0
    cfgd = out["config"]
    assert cfgd["rows"] == 2 * cfgd["rows_per_gpu"] and 0 < cfgd["ghost_slots_per_gpu_max"] < cfgd["rows_per_gpu"] // 10


def test_strong_allgather_three_ranks(gpu):
    out = _bench(3, ["--scaling", "strong"])
    assert out["n_gpus"] == 3 and out["scaling"] == "strong"
    assert out["parity"]["rows_over_1e-12"] == 0


@pytest.mark.parametrize("world", [1, 2])
def test_halo_cg_ranks_on_one_gpu(gpu, world):
    """HaloCG: conjugate gradients over the rank-local plans -- the direction vector travels through the
    halo exchange, the dot products through all_reduce -- plain and Jacobi-preconditioned with
    symmetric pair storage; checked by the true residual of the global system."""
    env = dict(os.environ)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "cg_worker.py")]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    assert "CG_OK" in p.stdout and p.stdout.count("CG_CASE") == 2, p.stdout[-2000:]
