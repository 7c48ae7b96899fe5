#!/usr/bin/env python3
"""Successive multiplies walking every partition in alternating directions (cfg.ell_alternate: a launch starts with what the
one before left in the Infinity Cache) against always first-to-last: one matrix, one reorder, two plans, arms alternating.
usage: python tools/alt_ab.py [--workload audikw_1-like] [--sym 1] [--rounds 3]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="audikw_1-like")
    ap.add_argument("--sym", type=int, default=1)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--iters", type=int, default=200)
    a = ap.parse_args()
    import bench as B
    import ehyb_spmv_gpu_amd as E
    from oracle import oracle as O

    gen, gargs, _ = B.WORKLOADS[a.workload]
    kw = dict(sym_pairs=1 if a.sym else 0, partitioner=B.partitioner_for(E, gen))
    cfg = E.make_config(**kw)
    m = E.Matrix.generate(gen, *gargs, cfg=cfg)
    x = O.x_glibc(m.n)
    y_ref = O.spmv_coo(m.n, m.I, m.J, m.V, x)
    scale = O.abs_rowsum(m.n, m.I, m.J, m.V, x)
    m.reorder(cfg)
    perm = m.reorder_list.copy()
    xp = E.vector_reorder(x, perm)
    plans = {"alternating": E.Plan(m, E.make_config(ell_alternate=1, **kw)), "first to last": E.Plan(m, E.make_config(ell_alternate=2, **kw))}
    dx, dy = E.DeviceBuffer(m.n).upload(xp), E.DeviceBuffer(m.n)
    for p in plans.values():
        p.tune(dx.ptr, dy.ptr)
    for r in range(a.rounds):
        for name, p in plans.items():
            t = p.bench(dx.ptr, dy.ptr, warmup=20, iters=a.iters)
            bad = int(O.check_tolerance(E.vector_recover(dy.download(), perm), y_ref, scale)[0])
            print(json.dumps({"workload": a.workload, "sym": a.sym, "arm": name, "round": r, "us": round(t["ms_total"] / a.iters * 1e3, 2),
                              "ell_us": round(t["ms_ell_avg"] * 1e3, 2), "er_us": round(t["ms_er_avg"] * 1e3, 2),
                              "gflops": round(2.0 * m.nnz / (t["ms_total"] / a.iters) / 1e6, 1), "rows_out_of_tolerance": bad}), flush=True)


if __name__ == "__main__":
    main()
