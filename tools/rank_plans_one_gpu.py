#!/usr/bin/env python3
"""Every rank's plan of an N-rank decomposition of ONE R-MAT, built and timed on ONE GPU, one rank after the other: what a rank's LOCAL
multiply costs (the whole of it, and the part that needs its own columns only) -- the figure the timeline model of the multi-GPU step
(tools/dist_volume_model.py) ASSUMES from a byte count and a rate, measured.  No collective runs: the ghost columns hold
whatever they hold (x = 1 everywhere), the ranks are threads of this process for the set-up (dist.ThreadRanks).

usage: python tools/rank_plans_one_gpu.py [--scale 24] [--ranks 8] [--exchange cover,halo] [--iters 50]
"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=int, default=24)
    ap.add_argument("--ranks", default="8")
    ap.add_argument("--exchange", default="cover,halo")
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--threads-per-rank", type=int, default=2)
    args = ap.parse_args()
    import ehyb_spmv_gpu_amd as E
    from ehyb_spmv_gpu_amd import dist as D

    lib = E.host._lib.load()
    edges = 1 << (args.scale + 3)
    for world in [int(v) for v in args.ranks.split(",")]:
        for exchange in args.exchange.split(","):
            cfg = E.make_config(partitioner=E.EHYB_PART_DEGREE, host_threads=args.threads_per_rank)
            tr = D.ThreadRanks(world)
            locals_, errs = [None] * world, []

            def build(rank):
                try:
                    m = E.Matrix.generate("rmat_block", args.scale, edges, 1, rank, world, 1 if exchange == "cover" else 0, cfg=cfg)
                    cuts = m.block_cuts
                    rp = m.row_idx.astype(np.int64)
                    a, b = int(rp[cuts[rank]]), int(rp[cuts[rank + 1]])
                    I, J, V = m.I[a:b].copy(), m.J[a:b].copy(), m.V[a:b].copy()
                    m.free()
                    locals_[rank] = D.RankLocalMatrix(I, J, V, cuts, rank, cfg, group=tr.group(rank), exchange=exchange, chunks=2, chunk_shares=[0.25, 0.75])
                except Exception as e:  # noqa: BLE001
                    errs.append((rank, repr(e)))
                    tr.barrier.abort()

            t0 = time.time()
            ts = [threading.Thread(target=build, args=(r,)) for r in range(world)]
            for t in ts:
                t.start()
            for t in ts:
                t.join()
            if errs:
                raise SystemExit(f"set-up failed: {errs}")
            t_setup = time.time() - t0
            rows = []
            for rank, L in enumerate(locals_):
                plan = L.plan()
                st = plan.stats
                n_x, n_y = L.n_loc + L.n_ext, L.n_loc + L.n_foreign
                dx, dy = E.DeviceBuffer(n_x).upload(np.ones(n_x)), E.DeviceBuffer(max(n_y, 1))
                r = plan.bench(dx.ptr, dy.ptr, warmup=5, iters=args.iters, per_kernel=False)
                whole_us = r["ms_total"] / args.iters * 1e3
                own_us = None
                if st["er_partials"] > 0 and st["er_inline"] == 0:
                    st0 = E.Stream()
                    for it in range(args.iters + 5):          # own columns only: segment 0 (+ the close of the foreign rows for the cover)
                        if it == 5:
                            st0.sync()
                            t1 = time.perf_counter()
                        plan.spmv_part(dx.ptr, dy.ptr, st0.ptr, 0, 1, 1 | (4 if L.cover else 0))
                    st0.sync()
                    own_us = (time.perf_counter() - t1) / args.iters * 1e6
                    st0.destroy()
                rows.append({"rank": rank, "rows": L.n_loc, "foreign_rows": L.n_foreign, "entries_multiplied": L.nnz, "ghost_columns": L.n_ghost,
                             "partial_sums_received": int(L.yrecv_counts.sum()), "format_MB": round(st["bytes_format"] / 1e6, 1), "er_partials": st["er_partials"],
                             "local_multiply_us": round(whole_us, 1), "own_columns_part_us": round(own_us, 1) if own_us else None,
                             "format_TBps": round(st["bytes_format"] / whole_us / 1e6, 2)})
                plan.destroy()
                dx.free(), dy.free()
            out = {"matrix": f"rmat-{args.scale}", "ranks": world, "exchange": exchange, "setup_s": round(t_setup, 1), "per_rank": rows,
                   "local_multiply_us_max": max(r["local_multiply_us"] for r in rows), "local_multiply_us_mean": round(float(np.mean([r["local_multiply_us"] for r in rows])), 1),
                   "format_MB_total": round(sum(r["format_MB"] for r in rows), 1),
                   "doubles_received_max": max(r["ghost_columns"] + r["partial_sums_received"] for r in rows)}
            print(json.dumps(out), flush=True)
            for L in locals_:
                L.m.free()
    lib.ehyb_dev_sync()


if __name__ == "__main__":
    main()
