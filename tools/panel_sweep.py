#!/usr/bin/env python3
"""Where the time of the panel residual goes at a given size: per-pass times (HIP events, each
pass alone) for a sweep of panel widths and row-block heights on ONE matrix and ONE permutation, with probe runs
that switch the partial stores of pass 1 off (EHYB probe 2: results wrong, timing only).

usage: python tools/panel_sweep.py [--workload rmat-24] [--panel-cols 8192,16384] [--block-rows 2048,4096,8192] [--iters 20]
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="rmat-24")
    ap.add_argument("--panel-cols", default="8192,16384")
    ap.add_argument("--block-rows", default="2048,4096,8192")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--probes", default="0,2")
    ap.add_argument("--units2", default="0", help="cfg.er_units2 values (row blocks aimed at; 0 = default 2048)")
    ap.add_argument("--prune-pct", default="110", help="cfg.prune_pct values: a window is given up when it costs more than this share of the panel form")
    args = ap.parse_args()
    import bench as B
    import ehyb_spmv_gpu_amd as E
    from oracle import oracle as O

    gen, gargs, _ = B.WORKLOADS[args.workload]
    part = B.partitioner_for(E, gen)
    cfg0 = E.make_config(partitioner=part)
    m = E.Matrix.generate(gen, *gargs, cfg=cfg0)
    n, nnz = m.n, m.nnz
    x = E.x_glibc(n)
    y_ref = O.spmv_coo(n, m.I, m.J, m.V, x)
    scale = O.abs_rowsum(n, m.I, m.J, m.V, x)
    m.reorder(cfg0)
    perm = m.reorder_list.copy()
    xd, yd = E.DeviceBuffer(n).upload(E.vector_reorder(x, perm)), E.DeviceBuffer(n)
    lib = E.host._lib.load()
    lib.ehyb_debug_panel_times.restype = C.c_int
    lib.ehyb_debug_panel_times.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    for pc in [int(v) for v in args.panel_cols.split(",")]:
      for br in [int(v) for v in args.block_rows.split(",")]:
       for u2 in [int(v) for v in args.units2.split(",")]:
        for pct in [int(v) for v in args.prune_pct.split(",")]:
            cfg = E.make_config(partitioner=part, fuse_er=2, er_mode=2, er_panel_cols=pc, er_block_rows=br, er_units2=u2, prune_pct=pct)
            t0 = time.time()
            plan = E.Plan(m, cfg)
            t_plan = time.time() - t0
            st = plan.stats
            r = plan.bench(xd.ptr, yd.ptr, warmup=5, iters=args.iters)
            bad, worst = O.check_tolerance(E.vector_recover(yd.download(), perm), y_ref, scale)
            out = {"workload": args.workload, "panel_cols": pc, "block_rows": br, "units2_aim": u2, "prune_pct": pct, "nnz_ell": st["nnz_ell"], "nnz_er": st["nnz_er"], "partials": st["er_partials"],
                   "items1": len(plan.array("pb_items1")) // 2, "units1": len(plan.array("pb_units1")) // 4, "units2": len(plan.array("pb_units2")) // 4,
                   "us_spmv": round(r["ms_total"] / args.iters * 1e3, 1), "us_ell": round(r["ms_ell_avg"] * 1e3, 1), "us_er": round(r["ms_er_avg"] * 1e3, 1),
                   "er_format_MB": round((st["bytes_format"] - st["bytes_format_ell"]) / 1e6, 1), "rows_over_tol": bad, "plan_s": round(t_plan, 1)}
            for probe in [int(v) for v in args.probes.split(",")]:
                a, b = C.c_double(), C.c_double()
                rc = lib.ehyb_debug_panel_times(plan.h, C.c_void_p(xd.ptr), C.c_void_p(yd.ptr), args.iters, probe, C.byref(a), C.byref(b))
                assert rc == 0, lib.ehyb_last_error()
                out[f"probe{probe}_us_scale"] = round(a.value * 1e3, 1)
                out[f"probe{probe}_us_reduce"] = round(b.value * 1e3, 1)
            print(json.dumps(out), flush=True)
            plan.destroy()


if __name__ == "__main__":
    main()
