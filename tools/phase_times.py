#!/usr/bin/env python3
"""Where the wall time of one small end-to-end case goes (generate, oracle, reorder, layout, upload,
multiply, destroy) -- a diagnostic for the test suite's per-case cost.  usage: python tools/phase_times.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ehyb_spmv_gpu_amd as E  # noqa: E402
from oracle import oracle as O  # noqa: E402

for rep in range(3):
    t = [time.perf_counter()]
    cfg = E.make_config(lds_doubles=2048, threads=256, sym_pairs=1)
    m = E.Matrix.generate("fem3d", 30000, 3, 22, 22, 13500, 1, 1, cfg=cfg); t.append(time.perf_counter())
    x = E.x_glibc(m.n)
    y = O.spmv_coo(m.n, m.I, m.J, m.V, x); s = O.abs_rowsum(m.n, m.I, m.J, m.V, x); t.append(time.perf_counter())
    m.reorder(cfg); t.append(time.perf_counter())
    plan = E.Plan(m, cfg, upload=False); t.append(time.perf_counter())
    plan.upload(); t.append(time.perf_counter())
    xp = E.vector_reorder(x, m.reorder_list)
    yp = plan.spmv_host(xp, iters=2); t.append(time.perf_counter())
    plan.destroy(); t.append(time.perf_counter())
    names = ["generate", "oracle", "reorder", "layout", "upload", "spmv_host", "destroy"]
    print("rep", rep, "  ".join(f"{n} {b - a:.3f}s" for n, a, b in zip(names, t, t[1:])))
