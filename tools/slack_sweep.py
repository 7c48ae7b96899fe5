#!/usr/bin/env python3
"""How far above the mean a partition of the symmetric-pair layout may grow (cfg.sym_slack_permille; one workgroup per
partition, so the largest one ends last): reorder + plan + tuned multiply per value, arms alternating in one process.
usage: python tools/slack_sweep.py [--workload audikw_1-like] [--slack 5,10,20,30,50] [--rounds 2]"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="audikw_1-like")
    ap.add_argument("--slack", default="5,10,20,30,50")
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--iters", type=int, default=200)
    a = ap.parse_args()
    import bench as B
    import ehyb_spmv_gpu_amd as E

    gen, gargs, _ = B.WORKLOADS[a.workload]
    plans = {}
    for s in [int(v) for v in a.slack.split(",")]:
        cfg = E.make_config(sym_pairs=1, sym_slack_permille=s)
        m = E.Matrix.generate(gen, *gargs, cfg=cfg)
        m.reorder(cfg)
        rows = np.diff(m.part_boundary[:m.c.nParts + 1].astype(np.int64))
        plan = E.Plan(m, cfg)
        dx, dy = E.DeviceBuffer(m.n).upload(E.x_glibc(m.n)), E.DeviceBuffer(m.n)
        plan.tune(dx.ptr, dy.ptr)
        plans[s] = (plan, dx, dy, {"slack_permille": s, "parts": int(m.c.nParts), "rows_max": int(rows.max()), "rows_mean": round(float(rows.mean()), 1),
                                   "empty_parts": int((rows == 0).sum()), "format_MB": round(plan.stats["bytes_format"] / 1e6, 2), "nnz": int(m.nnz)})
        m.free()
    for r in range(a.rounds):
        for s, (plan, dx, dy, info) in plans.items():
            t = plan.bench(dx.ptr, dy.ptr, warmup=20, iters=a.iters, per_kernel=False)["ms_total"] / a.iters
            print(json.dumps(dict(info, round=r, us=round(t * 1e3, 2), gflops=round(2.0 * info["nnz"] / t / 1e6, 1))), flush=True)


if __name__ == "__main__":
    main()
