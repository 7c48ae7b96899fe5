#!/usr/bin/env python3
"""What the ranks of `bench.py --gpus N` (strong scaling of one R-MAT) hold and exchange, WITHOUT GPUs: N gloo ranks on the
CPU build their rank-local matrices and host plans and report entries, the share in own columns, ghost columns, doubles
received per exchange step and peer, partial sums and format bytes of the local plan.  The numbers DESIGN.md 5's
estimate of the multi-GPU step is made from.

usage: python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29577 tools/dist_stats.py [--scale 24] [--chunks 2]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=int, default=24)
    ap.add_argument("--chunks", type=int, default=2)
    ap.add_argument("--chunk-shares", default="")
    ap.add_argument("--threads", type=int, default=1)
    args = ap.parse_args()
    import torch
    import torch.distributed as dist

    import ehyb_spmv_gpu_amd as E
    from ehyb_spmv_gpu_amd import dist as D

    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    cfg = E.make_config(partitioner=E.EHYB_PART_DEGREE, host_threads=args.threads)
    t0 = time.time()
    m = E.Matrix.generate("rmat_block", args.scale, 1 << (args.scale + 3), 1, rank, world, cfg=cfg)
    cuts = m.block_cuts
    rp = m.row_idx.astype(np.int64)
    a, b = int(rp[cuts[rank]]), int(rp[cuts[rank + 1]])
    I, J, V = m.I[a:b].copy(), m.J[a:b].copy(), m.V[a:b].copy()
    m.free()
    t_gen = time.time() - t0
    t0 = time.time()
    shares = [float(v) for v in args.chunk_shares.split(",")] if args.chunk_shares else None
    L = D.RankLocalMatrix(I, J, V, cuts, rank, cfg, chunks=args.chunks, chunk_shares=shares)
    t_local = time.time() - t0
    t0 = time.time()
    plan = L.plan(upload=False)
    t_plan = time.time() - t0
    st = plan.stats
    # entries per column segment (own, chunk 0, chunk 1, ...)
    segs = L.col_segs
    mm = L.m
    seg_of = np.searchsorted(segs, mm.J, side="right") - 1
    ent = np.bincount(seg_of, minlength=len(segs) - 1)
    out = {"rank": rank, "rows": L.n_loc, "nnz": int(len(V)), "nnz_own_cols": L.nnz_own_cols, "ghost_cols": L.n_ghost,
           "recv_doubles_step_peer": L.recv_counts.tolist(), "entries_by_segment": [int(e) for e in ent],
           "er_partials": st["er_partials"], "nnz_ell": st["nnz_ell"], "format_MB": round(st["bytes_format"] / 1e6, 1),
           "alg_MB": round((12 * len(V) + 16 * L.n_loc + 8 * (L.n_loc + L.n_ghost)) / 1e6, 1),
           "items_by_segment": np.diff(plan.array("pb_seg_item")).tolist() if st["er_partials"] else None,
           "gen_s": round(t_gen, 1), "local_s": round(t_local, 1), "plan_s": round(t_plan, 1)}
    gathered = [None] * world
    dist.all_gather_object(gathered, out)
    if rank == 0:
        for g in gathered:
            print(json.dumps(g), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
