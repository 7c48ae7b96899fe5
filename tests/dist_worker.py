"""Worker for tests/test_distributed_cpu.py: run under torch.distributed.run with the gloo
backend.  Exercises the N > 1 host logic -- two-level reorder, per-rank row blocks, plan per
block, the x exchange -- with the multiply itself replaced by the oracle's CPU walk of the
rank's layout (tests may use the oracle; the product never does)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import ehyb_spmv_gpu_amd as E  # noqa: E402
from ehyb_spmv_gpu_amd import dist as D  # noqa: E402
from oracle import oracle as O  # noqa: E402


def run_case(kind, args, cfg_kw, rank, world):
    cfg = E.make_config(n_top=world, **cfg_kw)
    m = E.Matrix.generate(kind, *args, cfg=cfg)
    n = m.n
    x = O.x_glibc(n)
    y_ref = O.spmv_coo(n, m.I, m.J, m.V, x)
    scale = O.abs_rowsum(n, m.I, m.J, m.V, x)
    m.reorder(cfg)
    perm = m.reorder_list.copy()
    xp = E.vector_reorder(x, perm)
    cuts = D.row_cuts(m, cfg, world)
    assert cuts[0] == 0 and cuts[-1] == n and all(a < b for a, b in zip(cuts, cuts[1:])), cuts
    r0, r1 = cuts[rank], cuts[rank + 1]
    plan = E.Plan(m, cfg, rows=(r0, r1), upload=False)
    # every rank starts with only its own x segment; the exchange must supply the rest
    x_full = torch.zeros(n, dtype=torch.float64)
    x_full[r0:r1] = torch.from_numpy(xp[r0:r1])
    D.exchange_segments(x_full, cuts, rank)
    assert np.array_equal(x_full.numpy(), xp), "exchange did not reproduce the full x"
    # ELL windows of a block only reference the block's own segment (what makes overlap legal)
    hc = plan.array("halo_cols")
    assert len(hc) == 0 or (hc.min() >= r0 and hc.max() < r1)
    y_loc, written = O.walk_plan(plan, x_full.numpy())
    assert written[r0:r1].min() == 1 and written.sum() == r1 - r0
    y_full = torch.zeros(n, dtype=torch.float64)
    y_full[r0:r1] = torch.from_numpy(y_loc[r0:r1])
    D.exchange_segments(y_full, cuts, rank)  # iterating x <- y is the same exchange
    y = E.vector_recover(y_full.numpy(), perm)
    bad, worst = O.check_tolerance(y, y_ref, scale)
    st = plan.stats
    # entry balance of the blocks (one GPU each)
    tot = torch.tensor([float(st["nnz"])], dtype=torch.float64)
    mx = tot.clone()
    dist.all_reduce(tot)
    dist.all_reduce(mx, op=dist.ReduceOp.MAX)
    assert mx.item() <= tot.item() / world * 1.15, (mx.item(), tot.item())
    if rank == 0:
        print(f"DIST_CASE {kind} world={world} cuts={cuts} bad={bad} worst={worst:.2e} "
              f"remote_frac={st['nnz_er'] / max(1, st['nnz']):.3f}", flush=True)
    return bad


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    bad = 0
    bad += run_case("fem3d", (12000, 3, 16, 16, 13500, 1, 3), dict(lds_doubles=512), rank, world)          # ragged blocks
    bad += run_case("rmat", (12, 1 << 15, 2), dict(lds_doubles=256, window_mode=1), rank, world)            # heavy residual
    bad += run_case("banded", (1 << 13, 32, 1024), dict(lds_doubles=1024, window_mode=1, partitioner=1), rank, world)  # equal blocks
    dist.barrier()
    if rank == 0:
        print("DIST_OK" if bad == 0 else f"DIST_FAIL {bad}", flush=True)
    dist.destroy_process_group()
    sys.exit(0 if bad == 0 else 1)


if __name__ == "__main__":
    main()
