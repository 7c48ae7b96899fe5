#!/usr/bin/env python3
"""Tuning sweep on one GPU: partition once per window size, then time plan variants.

usage: python tools/sweep.py [--workload audikw_1-like] [--lds 10240,20480] [--threads 512,1024]
                             [--variants 1,2] [--items 2,4,8] [--iters 100]
Prints one line per configuration (interleaved A/B in one process, as the guide asks).
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def ints(s):
    return [int(v) for v in s.split(",") if v]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="audikw_1-like")
    ap.add_argument("--lds", default="10240")
    ap.add_argument("--rows-frac", default="550", help="part_rows as per-mille of lds (comma list)")
    ap.add_argument("--threads", default="1024")
    ap.add_argument("--variants", default="1", help="ell_variant values: 1 LDS slab counter (default), 3 static round-robin")
    ap.add_argument("--items", default="2")
    ap.add_argument("--mode", default="2")
    ap.add_argument("--sharing", default="1", help="col_sharing values: 1 on, 2 off")
    ap.add_argument("--fuse", default="0", help="fuse_er values: 0 automatic, 1 residual inline in the ELL launch, 2 own launch ('sh' column prints sharing*10+fuse)")
    ap.add_argument("--capsplit", type=int, default=1, help="cap_split of the reorder step: 1 on, 2 off")
    ap.add_argument("--iters", type=int, default=100)
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--sym", type=int, default=0, help="sym_pairs: 1 symmetric pair storage (changes the partitioning), 2 off")
    args = ap.parse_args()

    import numpy as np

    import bench as B
    import ehyb_spmv_gpu_amd as E

    gen, gargs, _ = B.WORKLOADS[args.workload]
    print(f"# workload {args.workload}", flush=True)
    for mode in ints(args.mode):
        for lds in ints(args.lds):
            for frac in ints(args.rows_frac):
                part_rows = max(64, lds * frac // 1000 // 64 * 64)
                cfg0 = E.make_config(lds_doubles=lds, part_rows=part_rows, window_mode=mode, cap_split=args.capsplit, sym_pairs=args.sym)
                t0 = time.time()
                m = E.Matrix.generate(gen, *gargs, cfg=cfg0)
                n, nnz = m.n, m.nnz
                x = E.x_glibc(n)
                m.reorder(cfg0)
                xp = E.vector_reorder(x, m.reorder_list)
                print(f"# mode {mode} lds {lds} part_rows {part_rows}: parts {m.c.nParts}, prep {time.time() - t0:.1f}s", flush=True)
                dx, dy = E.DeviceBuffer(n).upload(xp), E.DeviceBuffer(n)
                plans = []
                for threads in ints(args.threads):
                    for var in ints(args.variants):
                        for ipc in ints(args.items):
                            for sh in ints(args.sharing):
                                for fu in ints(args.fuse):
                                    cfg = E.make_config(lds_doubles=lds, part_rows=part_rows, window_mode=mode, threads=threads,
                                                        ell_variant=var, items_per_cu=ipc, col_sharing=sh, fuse_er=fu, sym_pairs=args.sym)
                                    plans.append(((threads, var, ipc, sh * 10 + fu), E.Plan(m, cfg)))
                ref = None
                best = {}
                for rnd in range(args.rounds):
                    for key, plan in plans:
                        r = plan.bench(dx.ptr, dy.ptr, warmup=5, iters=args.iters)
                        y = dy.download()
                        if ref is None:
                            ref = y
                        err = float(np.max(np.abs(y - ref)))
                        cur = best.get(key)
                        if cur is None or r["ms_ell_avg"] < cur["ms_ell_avg"]:
                            best[key] = dict(r, err=err)
                for key, plan in plans:
                    st = plan.stats
                    r = best[key]
                    t = r["ms_total"] / args.iters
                    print(f"mode {mode} lds {lds:6d} rows {part_rows:6d} thr {key[0]:5d} var {key[1]} ipc {key[2]:2d} sh {key[3]} items {st['n_items']:5d} "
                          f"ell {st['nnz_ell'] / nnz * 100:6.2f}% pad {st['ell_padding'] / st['size_block_ell'] * 100:5.2f}% "
                          f"| spmv {t * 1e3:8.1f} us  ell {r['ms_ell_avg'] * 1e3:8.1f} us  er {r['ms_er_avg'] * 1e3:7.1f} us "
                          f"| {2 * nnz / t / 1e6:8.1f} GFLOP/s  alg {st['bytes_alg'] / t / 1e6:7.1f} GB/s  maxdiff {r['err']:.1e}",
                          flush=True)
                for _, plan in plans:
                    plan.destroy()
                m.free()


if __name__ == "__main__":
    main()
