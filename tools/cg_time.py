#!/usr/bin/env python3
"""Time per iteration of the device-resident CG (ehyb_pcg) on a bench workload made positive
definite (the generator's diagonal is replaced by row sum of |a_ij| + 1), with the iteration
replayed from a hipGraph and with plain launches (cfg.graphs = 2), beside the SpMV alone.

usage: python tools/cg_time.py [--workload audikw_1-like] [--iters 20,120] [--sym-pairs 1]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="audikw_1-like")
    ap.add_argument("--iters", default="20,120")
    ap.add_argument("--sym-pairs", type=int, default=1)
    ap.add_argument("--jacobi", action="store_true")
    args = ap.parse_args()
    import bench as B
    import ehyb_spmv_gpu_amd as E

    gen, gargs, _ = B.WORKLOADS[args.workload]
    cfg = E.make_config(sym_pairs=args.sym_pairs)
    m = E.Matrix.generate(gen, *gargs, cfg=cfg)
    I, J, V = m.I, m.J, m.V
    off = np.bincount(I, weights=np.abs(V) * (I != J), minlength=m.n)
    V[I == J] = (off + 1.0)[I[I == J]]          # strictly diagonally dominant, still symmetric
    m.reorder(cfg)
    I, J, V = m.I, m.J, m.V
    diag = np.zeros(m.n)
    diag[I[I == J]] = V[I == J]
    plan = E.Plan(m, cfg)
    st = plan.stats
    b = np.ones(m.n)
    dx, dy = E.DeviceBuffer(m.n).upload(b), E.DeviceBuffer(m.n)
    spmv_us = plan.bench(dx.ptr, dy.ptr, warmup=20, iters=200)["ms_total"] * 1e3 / 200
    print(f"# {args.workload}: n={m.n} nnz={m.nnz} sym_pairs={st['sym_pairs']}; SpMV alone {spmv_us:.1f} us")
    lo, hi = [int(v) for v in args.iters.split(",")]
    inv = 1.0 / diag if args.jacobi else None
    # arms: graph replay with p.q left by the multiply (the default), graph replay with the separate dot kernel, plain launches
    plans = {"1": plan, "1, separate dot kernel": E.Plan(m, E.make_config(sym_pairs=args.sym_pairs, cg_fused_dot=2)),
             "0": E.Plan(m, E.make_config(sym_pairs=args.sym_pairs, graphs=2))}
    # (wall clock of a long solve minus a short one; three rounds, arms alternating, the smallest difference per arm: one-off
    # costs of a call -- workspace, capture, instantiation -- land in either and do not cancel exactly)
    best = {k: None for k in plans}
    last_rel = {}
    for k, plan in plans.items():
        plan.cg(b, max_iter=10, rtol=0.0, check_every=10, inv_diag=inv)  # warm
    for _ in range(3):
        for graph, plan in plans.items():
            t = {}
            for it in (lo, hi):
                t0 = time.perf_counter()
                _, done, rel = plan.cg(b, max_iter=it, rtol=0.0, check_every=max(20, it // 4), inv_diag=inv)
                t[it] = time.perf_counter() - t0
                assert done == it, (done, it)
            per = (t[hi] - t[lo]) / (hi - lo) * 1e6
            best[graph] = per if best[graph] is None else min(best[graph], per)
            last_rel[graph] = rel
    for graph, per in best.items():
        print(f"graph={graph}: {per:7.1f} us per CG iteration ({per - spmv_us:6.1f} us beyond the SpMV); "
              f"rel. residual after {hi}: {last_rel[graph]:.2e}")


if __name__ == "__main__":
    main()
