#!/usr/bin/env python3
"""Same-box A/B of ehyb_plan_tune (the heaviest work items on the XCDs measured fastest): time per SpMV of one plan before
and after the call, and again after a second call; parity after.   usage: python tools/tune_ab.py [--workloads a,b] [--iters 300]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workloads", default="audikw_1-like,audikw_1-mesh,audikw_1-graded,kkt3d-110,small")
    ap.add_argument("--iters", type=int, default=300)
    ap.add_argument("--sym", default="1,0")
    args = ap.parse_args()
    import bench as B
    import ehyb_spmv_gpu_amd as E
    from oracle import oracle as O

    for wl in args.workloads.split(","):
        gen, gargs, _ = B.WORKLOADS[wl]
        for sym in [int(v) for v in args.sym.split(",")]:
            cfg = E.make_config(sym_pairs=sym, partitioner=B.partitioner_for(E, gen))
            m = E.Matrix.generate(gen, *gargs, cfg=cfg)
            n = m.n
            x = E.x_glibc(n)
            y_ref = O.spmv_coo(n, m.I, m.J, m.V, x)
            scale = O.abs_rowsum(n, m.I, m.J, m.V, x)
            m.reorder(cfg)
            perm = m.reorder_list.copy()
            plan = E.Plan(m, cfg)
            dx, dy = E.DeviceBuffer(n).upload(E.vector_reorder(x, perm)), E.DeviceBuffer(n)
            us = []
            spans = []
            for rnd in range(3):
                r = plan.bench(dx.ptr, dy.ptr, warmup=20, iters=args.iters, per_kernel=False)
                us.append(round(r["ms_total"] / args.iters * 1e3, 2))
                if rnd < 2:
                    spans.append([round(v, 1) for v in plan.tune(dx.ptr, dy.ptr)])
            plan.spmv(dx.ptr, dy.ptr)
            E.host._lib.load().ehyb_dev_sync()
            bad, worst = O.check_tolerance(E.vector_recover(dy.download(), perm), y_ref, scale)
            st = plan.stats
            print(json.dumps({"workload": wl, "sym": sym, "items": st["n_items"], "us_untuned": us[0], "us_tuned": us[1], "us_tuned_twice": us[2],
                              "stamped_spans_us": spans, "rows_over_tol": bad}), flush=True)
            plan.destroy()
            m.free()


if __name__ == "__main__":
    main()
