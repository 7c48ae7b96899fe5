#!/usr/bin/env python3
"""Symmetric pair storage against plain storage on small fem3d matrices: where the one-workgroup-per-partition
form stops paying.  usage: python tools/sym_crossover.py"""
import sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ehyb_spmv_gpu_amd as E
for n, nx in ((21000, 20), (42000, 24), (84000, 30)):
    res = {}
    for sym in (1, 0):
        cfg = E.make_config(sym_pairs=sym)
        m = E.Matrix.generate("fem3d", n, 3, nx, nx, 13500, 1, 3, cfg=cfg)
        m.reorder(cfg)
        plan = E.Plan(m, cfg)
        dx, dy = E.DeviceBuffer(m.n).upload(E.x_glibc(m.n)), E.DeviceBuffer(m.n)
        t = min(plan.bench(dx.ptr, dy.ptr, warmup=20, iters=1000, per_kernel=False)["ms_total"] for _ in range(3))  # ms per 1000 = us each
        res[sym] = (t, plan.stats["n_items"])
    print(f"rows {n} nnz {m.nnz}: sym {res[1][0]:.2f} us ({res[1][1]} items)  plain {res[0][0]:.2f} us ({res[0][1]} items)")
