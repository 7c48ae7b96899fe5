#!/usr/bin/env python3
"""A/B of the residual forms on one GPU box: CSR segments (er_mode 1) vs panel form (er_mode 2), same
matrix, same permutation; whole SpMV and per-launch times, parity of each against the CPU oracle.

usage: python tools/er_ab.py [--workloads rmat-22,kkt3d-110c] [--iters 100] [--panel-cols 8192,16384]
  a workload name with a trailing "c" forces contiguous partitions (a residual-heavy structured case).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workloads", default="rmat-22")
    ap.add_argument("--iters", type=int, default=100)
    ap.add_argument("--panel-cols", default="8192")
    ap.add_argument("--block-rows", default="8192")
    args = ap.parse_args()
    import bench as B
    import ehyb_spmv_gpu_amd as E
    from oracle import oracle as O

    for wl in args.workloads.split(","):
        contiguous = wl.endswith("c") and wl[:-1] in B.WORKLOADS
        name = wl[:-1] if contiguous else wl
        gen, gargs, _ = B.WORKLOADS[name]
        part = E.EHYB_PART_CONTIGUOUS if contiguous else B.partitioner_for(E, gen)
        cfg0 = E.make_config(partitioner=part, er_mode=1)
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
        arms = [("csr", dict(er_mode=1))]
        for pc in args.panel_cols.split(","):
            for br in args.block_rows.split(","):
                arms.append((f"panel{pc}x{br}", dict(er_mode=2, er_panel_cols=int(pc), er_block_rows=int(br))))
                arms.append((f"panel{pc}x{br}-windows-kept", dict(er_mode=2, er_panel_cols=int(pc), er_block_rows=int(br), ell_prune=2)))
        for tag, kw in arms:
            cfg = E.make_config(partitioner=part, fuse_er=2, **kw)
            t0 = time.time()
            plan = E.Plan(m, cfg)
            t_plan = time.time() - t0
            r = plan.bench(xd.ptr, yd.ptr, warmup=10, iters=args.iters)
            bad, worst = O.check_tolerance(E.vector_recover(yd.download(), perm), y_ref, scale)
            st = plan.stats
            ms = r["ms_total"] / args.iters
            print(json.dumps({"workload": wl, "arm": tag, "rows": n, "nnz": nnz, "nnz_ell": st["nnz_ell"], "nnz_er": st["nnz_er"], "er_partials": st["er_partials"],
                              "us_spmv": round(ms * 1e3, 2), "us_ell": round(r["ms_ell_avg"] * 1e3, 2), "us_er": round(r["ms_er_avg"] * 1e3, 2),
                              "GFLOPs": round(2.0 * nnz / ms / 1e6, 1), "er_format_bytes": st["bytes_format"] - st["bytes_format_ell"],
                              "er_GBps": round((st["bytes_format"] - st["bytes_format_ell"]) / max(r["ms_er_avg"], 1e-9) / 1e6, 1),
                              "rows_over_tol": bad, "worst": worst, "plan_s": round(t_plan, 1), "reorder_s": round(t_re, 1)}), flush=True)
            plan.destroy()
        m.free()


if __name__ == "__main__":
    main()
