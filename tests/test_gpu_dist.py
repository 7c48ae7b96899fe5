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


def _bench(world, extra, workload="small", launcher=True):
    """launcher=True: as the driver starts it (torch.distributed.run); False: `python bench.py --gpus N`
    invoked plainly, which must start its ranks itself as a child process."""
    env = dict(os.environ, EHYB_BENCH_ONE_DEVICE="1", EHYB_BENCH_BACKEND="gloo")
    env.pop("WORLD_SIZE", None)
    tail = [os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "10", "--warmup", "2", "--workload", workload] + extra
    if launcher:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
               "--master-addr", "127.0.0.1", "--master-port", str(_free_port())] + tail
    else:
        cmd = [sys.executable] + tail
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    lines = [ln for ln in p.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def test_weak_halo_exchange_two_ranks(gpu):
    out = _bench(2, ["--scaling", "weak"])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak"
    assert out["parity"]["rows_over_1e-12"] == 0
    cfgd = out["config"]
    assert cfgd["rows"] == 2 * cfgd["rows_per_gpu"] and 0 < cfgd["ghost_slots_per_gpu_max"] < cfgd["rows_per_gpu"] // 10


def test_strong_halo_rmat_two_ranks_self_launched(gpu):
    """The default N > 1 mode (strong scaling of one R-MAT, halo exchange) started plainly as
    `python bench.py --gpus 2`: the bench spawns its own ranks before touching the GPU."""
    out = _bench(2, [], workload="rmat-18", launcher=False)
    assert out["n_gpus"] == 2 and out["scaling"] == "strong"
    assert out["parity"]["rows_over_1e-12"] == 0
    c = out["config"]
    assert "all_to_all" in c["exchange"] and 0 < c["ghost_columns_per_gpu_max"] < c["rows"]
    assert out["local_multiply_ms_max_over_ranks"] > 0
    # the default exchange of the R-MAT workloads is "cover": hub columns as x entries, the rest of every block as partial sums
    cov = c["cover"]
    assert c["exchange"].startswith("cover") and cov["partial_sums_all_gpus"] > 0 and cov["entries_handed_to_the_column_owner"] > 0
    assert cov["ghost_columns_all_gpus"] + cov["partial_sums_all_gpus"] == c["exchange_doubles_received_all_gpus"]


def test_strong_allgather_three_ranks(gpu):
    """--exchange allgather: the padded x segments through one all-gather per step, read in place by
    the residual phase (fem3d matrix sharded by rows, symmetric pair storage inside every block)."""
    out = _bench(3, ["--exchange", "allgather"])
    assert out["n_gpus"] == 3 and out["scaling"] == "strong"
    assert out["parity"]["rows_over_1e-12"] == 0
    assert "all_gather_into_tensor" in out["config"]["exchange"]
    # the row blocks come from a graph partition: a GPU's rows reference only a fraction of the others' columns
    assert out["config"]["ghost_columns_per_gpu_max"] < out["config"]["rows"] // 6


def test_world_size_mismatch_is_refused(gpu):
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2"], capture_output=True, text=True,
                       timeout=120, env=env, cwd=ROOT)
    assert p.returncode != 0 and "WORLD_SIZE=2" in p.stderr


@pytest.mark.parametrize("world", [2, 3])
def test_pipelined_step_equals_plain_step(gpu, world):
    """The chunked, pipelined exchange step (own-column panels while chunk 0 travels, chunk k's panels while chunk k+1
    travels, one C call per part) against the plain step (every chunk, then one multiply) on the same plan: R-MAT 2^18 in
    panel form with 3 chunks (all_to_all) and 2 chunks (send/recv pairs), CSR residual, and a FEM matrix with windows --
    equal to rounding of the atomics' order (bit for bit where the kernels are deterministic), all equal to the oracle."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "pipeline_worker.py")]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=dict(os.environ), cwd=ROOT)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    assert "PIPE_OK" in p.stdout and p.stdout.count("PIPE_CASE") == 6, p.stdout[-2000:]


def test_strong_bench_line_carries_the_step_anatomy(gpu):
    """bench.py --gpus 2 (strong, halo, 3 chunks): the JSON line says how much of the local multiply can run before the
    first chunk lands, what the host spends per step, and who receives how much from whom per exchange step."""
    out = _bench(2, ["--exchange", "halo", "--chunks", "3", "--chunk-shares", "0.2,0.3,0.5"], workload="rmat-18")
    assert out["parity"]["rows_over_1e-12"] == 0
    c = out["config"]
    assert c["exchange_steps"] == 3 and c["pipelined"] is True and c["exchange_mode"] == "a2a"
    vol = c["recv_doubles_by_rank_step_peer"]
    assert len(vol) == 2 and all(len(r) == 3 and all(len(k) == 2 for k in r) for r in vol)
    assert sum(sum(sum(k) for k in r) for r in vol) == c["exchange_doubles_received_all_gpus"]
    assert vol[0][0][0] == 0 and vol[1][0][1] == 0          # nobody receives from itself
    assert 0 < out["phase1_share_of_local_ms"] < 1 and out["host_us_per_step"] > 0
    assert 0 < out["own_columns_share_of_entries"] < 1
    # the line carries its own N = 1 point: the same matrix on rank 0's GPU alone, in the same run
    one = out["single_gpu_same_matrix"]
    assert one["value"] > 0 and one["nnz"] == c["nnz"] and out["speedup_vs_single_gpu_same_run"] > 0


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
