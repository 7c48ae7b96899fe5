#!/usr/bin/env python3
"""Plain storage: rows per partition (cfg.part_rows) against the time per SpMV -- fewer, larger partitions
stage fewer halo columns, but below 256 partitions several work items re-stage the same window.
usage: python tools/part_rows_sweep.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ehyb_spmv_gpu_amd as E  # noqa: E402

CASES = [("fem3d", (196608, 3, 42, 42, 13500, 1, 1)), ("fem3d", (393216, 3, 52, 52, 13500, 1, 1)), ("fem3d", (943695, 3, 68, 68, 13500, 1, 1)),
         ("kkt3d", (110,))]
SIZES = (11264, 8192, 5632, 3712, 2048, 1024)
if len(sys.argv) > 1 and sys.argv[1] == "mid":   # the range where the direct shape hands over to the windows
    CASES = [("fem3d", (n, 3, 40, 40, 13500, 1, 1)) for n in (98304, 131070, 163839, 196608, 262143)]
    SIZES = (11264, 8192, 4096, 2048)
for kind, args in CASES:
    for pr in SIZES:
        kw = dict(direct=2)
        if pr:
            kw["part_rows"] = pr
        cfg = E.make_config(**kw)
        m = E.Matrix.generate(kind, *args, cfg=cfg)
        x = E.x_glibc(m.n)
        m.reorder(cfg)
        plan = E.Plan(m, cfg)
        st = plan.stats
        dx, dy = E.DeviceBuffer(m.n).upload(E.vector_reorder(x, m.reorder_list)), E.DeviceBuffer(m.n)
        r = plan.bench(dx.ptr, dy.ptr, warmup=20, iters=300, per_kernel=False)
        print(json.dumps({"matrix": f"{kind}{args[0]}", "part_rows": pr or int(cfg.part_rows), "parts": st["n_parts"], "items": st["n_items"],
                          "us": round(r["ms_total"] / 300 * 1e3, 2), "window_loads": st["window_loads"], "halo_cols": st["halo_cols"],
                          "nnz_er": st["nnz_er"], "fmt_MB": round(st["bytes_format"] / 1e6, 1)}), flush=True)
        plan.destroy()
        m.free()
