"""Where the pre-step of a power-law matrix spends its time: reorder, then the plan built (a) on the host
(ehyb_plan_create_host + ehyb_plan_upload, cfg.symbolic = 1) and (b) with the panel form left to the device
(ehyb_plan_create, cfg.symbolic = 2).  One JSON line per arm; --check multiplies with both plans against the oracle.
    python tools/plan_time.py --scale 24 --edges 27 [--verbose 2] [--check]"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ehyb_spmv_gpu_amd as E  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=int, default=22)
    ap.add_argument("--edges", type=int, default=0, help="log2 of the edge samples (default scale + 3)")
    ap.add_argument("--verbose", type=int, default=0)
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--arms", default="device,host,device")
    a = ap.parse_args()
    edges = a.edges or a.scale + 3
    base = dict(partitioner=E.EHYB_PART_DEGREE, verbose=a.verbose)
    cfg0 = E.make_config(**base)
    t = time.time()
    m = E.Matrix.generate("rmat", a.scale, 1 << edges, 1, cfg=cfg0)
    t_gen = time.time() - t
    if a.check:
        from oracle import oracle as O
        x = O.x_glibc(m.n)
        y_ref = O.spmv_coo(m.n, m.I, m.J, m.V, x)
        scale = O.abs_rowsum(m.n, m.I, m.J, m.V, x)
    t = time.time()
    m.reorder(cfg0)
    t_reorder = time.time() - t
    print(json.dumps({"matrix": f"rmat-{a.scale}", "rows": m.n, "entries": m.nnz, "generate_s": round(t_gen, 3), "reorder_s": round(t_reorder, 3),
                      "host_threads": cfg0.host_threads}), flush=True)
    for arm in a.arms.split(","):
        cfg = E.make_config(symbolic=1 if arm == "host" else 2, **base)
        t = time.time()
        plan = E.Plan(m, cfg)
        dt = time.time() - t
        st = plan.stats
        rec = {"arm": arm, "plan_create_s": round(dt, 3), "er_partials": st["er_partials"], "er_segments": st["er_segments"], "bytes_format": st["bytes_format"]}
        if a.check:
            perm = m.reorder_list
            y = E.vector_recover(plan.spmv_host(E.vector_reorder(x, perm)), perm)
            rec["rows_out_of_tolerance"] = int(O.check_tolerance(y, y_ref, scale)[0])
        print(json.dumps(rec), flush=True)
        plan.destroy()


if __name__ == "__main__":
    main()
