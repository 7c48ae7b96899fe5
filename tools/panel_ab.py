#!/usr/bin/env python3
"""Same-box A/B of the panel-form residual: row/column order (contiguous blocks of the given
numbering vs blocks of the degree order), panel width, how pass 1 adds up the pieces (register scan vs LDS words) and
whether the units of a panel share an XCD.  One matrix per order, one plan per arm, parity of every arm against the
CPU oracle; per-pass times from ehyb_debug_panel_times (each pass alone between HIP events).

usage: python tools/panel_ab.py [--workloads rmat-22] [--iters 50] [--orders 1,4] [--panel-cols 8192,16384]
                                [--sums 1,2] [--xcd 1,2] [--units1 0] [--block-rows 2048]
"""
import argparse
import ctypes as C
import itertools
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def ints(s):
    return [int(v) for v in s.split(",")]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workloads", default="rmat-22")
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--orders", default="1,4", help="cfg.partitioner values: 1 = contiguous blocks, 4 = blocks of the degree order")
    ap.add_argument("--panel-cols", default="8192,16384")
    ap.add_argument("--sums", default="1,2", help="cfg.er_sums: 1 = register scan (DPP), 2 = LDS words")
    ap.add_argument("--xcd", default="1,2", help="cfg.xcd_map: 1 = units of a panel on one XCD, 2 = blockIdx order")
    ap.add_argument("--units1", default="0")
    ap.add_argument("--block-rows", default="2048")
    ap.add_argument("--queue", default="2", help="cfg.er_queue: 1 = per-XCD work queues with stealing, 2 = one workgroup per item")
    ap.add_argument("--threads1", default="0", help="cfg.er_panel_threads: pass-1 workgroup size (0 automatic, 512, 1024)")
    args = ap.parse_args()
    import bench as B
    import ehyb_spmv_gpu_amd as E
    from oracle import oracle as O

    lib = E.host._lib.load()
    lib.ehyb_debug_panel_times.restype = C.c_int
    lib.ehyb_debug_panel_times.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    for wl in args.workloads.split(","):
        gen, gargs, _ = B.WORKLOADS[wl]
        for part in ints(args.orders):
            cfg0 = E.make_config(partitioner=part)
            m = E.Matrix.generate(gen, *gargs, cfg=cfg0)
            n, nnz = m.n, m.nnz
            x = E.x_glibc(n)
            y_ref = O.spmv_coo(n, m.I, m.J, m.V, x)
            scale = O.abs_rowsum(n, m.I, m.J, m.V, x)
            t0 = time.time()
            m.reorder(cfg0)
            t_re = time.time() - t0
            perm = m.reorder_list.copy()
            xd, yd = E.DeviceBuffer(n).upload(E.vector_reorder(x, perm)), E.DeviceBuffer(n)
            for pc, br, u1, th1 in itertools.product(ints(args.panel_cols), ints(args.block_rows), ints(args.units1), ints(args.threads1)):
                first = True
                for sums, xcd, queue in itertools.product(ints(args.sums), ints(args.xcd), ints(args.queue)):
                    cfg = E.make_config(partitioner=part, fuse_er=2, er_panel_cols=pc, er_block_rows=br, er_units1=u1, er_sums=sums, xcd_map=xcd, er_panel_threads=th1, er_queue=queue)
                    t0 = time.time()
                    plan = E.Plan(m, cfg)
                    t_plan = time.time() - t0
                    st = plan.stats
                    r = plan.bench(xd.ptr, yd.ptr, warmup=10, iters=args.iters)
                    bad, worst = O.check_tolerance(E.vector_recover(yd.download(), perm), y_ref, scale)
                    ms = r["ms_total"] / args.iters
                    out = {"workload": wl, "order": part, "panel_cols": pc, "block_rows": br, "units1_aim": u1, "threads1": th1, "er_sums": sums, "xcd_map": xcd, "er_queue": queue,
                           "nnz": nnz, "nnz_ell": st["nnz_ell"], "partials": st["er_partials"],
                           "items1": len(plan.array("pb_items1")) // 2 if first else None, "units1": len(plan.array("pb_units1")) // 4 if first else None, "units2": len(plan.array("pb_units2")) // 4 if first else None,
                           "us_spmv": round(ms * 1e3, 1), "GFLOPs": round(2.0 * nnz / ms / 1e6, 1), "us_ell": round(r["ms_ell_avg"] * 1e3, 1),
                           "us_er": round(r["ms_er_avg"] * 1e3, 1), "format_MB": round(st["bytes_format"] / 1e6, 1), "alg_MB": round(st["bytes_alg"] / 1e6, 1),
                           "rows_over_tol": bad, "worst": float(f"{worst:.2e}"), "plan_s": round(t_plan, 1), "reorder_s": round(t_re, 1)}
                    if st["er_partials"] > 0:
                        for probe in (0, 2):
                            a, b = C.c_double(), C.c_double()
                            rc = lib.ehyb_debug_panel_times(plan.h, C.c_void_p(xd.ptr), C.c_void_p(yd.ptr), 20, probe, C.byref(a), C.byref(b))
                            assert rc == 0, lib.ehyb_last_error()
                            out["us_scale" if probe == 0 else "us_scale_no_stores"] = round(a.value * 1e3, 1)
                            if probe == 0:
                                out["us_reduce"] = round(b.value * 1e3, 1)
                    print(json.dumps(out), flush=True)
                    plan.destroy()
                    first = False
            m.free()


if __name__ == "__main__":
    main()
