#!/usr/bin/env python3
"""Where the shapes cross over (fem3d, 3 unknowns per node): direct shape (no window, one small launch),
window shape with every entry stored, symmetric pair storage -- microseconds per SpMV (graph replay).
Behind EHYB_DIRECT_MAX_ROWS / EHYB_SYM_MIN_ROWS in include/ehyb.h."""
import sys, json
sys.path.insert(0, ".")
import ehyb_spmv_gpu_amd as E
for rows3 in (8192, 10922, 13653, 16384, 21845, 27306, 32768, 43690, 65536):
    n = rows3 * 3
    res = {}
    for d in (1, 2, 3):
        cfg = E.make_config(direct=d) if d < 3 else E.make_config(sym_pairs=1)
        m = E.Matrix.generate("fem3d", n, 3, 40, 40, 13500, 1, 1, cfg=cfg); x = E.x_glibc(m.n); m.reorder(cfg)
        plan = E.Plan(m, cfg)
        dx, dy = E.DeviceBuffer(m.n).upload(E.vector_reorder(x, m.reorder_list)), E.DeviceBuffer(m.n)
        r = plan.bench(dx.ptr, dy.ptr, warmup=50, iters=1000, per_kernel=False)
        res[d] = round(r["ms_total"], 3)
        nnz = m.nnz
        plan.destroy(); m.free()
    print(json.dumps({"rows": n, "nnz": nnz, "us_direct": res[1], "us_window": res[2], "us_sym_pairs": res[3]}))
