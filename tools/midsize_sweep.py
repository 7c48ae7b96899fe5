#!/usr/bin/env python3
"""Mid-size matrices (100 k - 400 k rows: fewer full-size partitions than CUs; the range the reference's
kernelCachedBlockedELL_small exists for, kernel.cu:197-284): partitions asked for x workgroup size x storage against the
time per SpMV, on two generators (FEM 3 unknowns per node, KKT saddle point).  What ehyb_sizing's mid-size rules are
derived from.

usage: python tools/midsize_sweep.py [--cases fem3d:120000,fem3d:196608,fem3d:393216,kkt3d:40,kkt3d:46,kkt3d:58]
                                     [--parts 0,128,256,384,512,768,1024] [--threads 512,1024] [--sym 1,0] [--iters 300]
  parts = 0: what ehyb_sizing chooses by itself (the default path).
"""
import argparse
import json
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ehyb_spmv_gpu_amd as E  # noqa: E402

FEM = {120000: (35, 35), 196608: (42, 42), 393216: (52, 52), 262143: (46, 46), 98304: (32, 32)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", default="fem3d:120000,fem3d:196608,fem3d:393216,kkt3d:40,kkt3d:46,kkt3d:58")
    ap.add_argument("--parts", default="0,128,256,384,512,768,1024")
    ap.add_argument("--threads", default="512,1024")
    ap.add_argument("--sym", default="1,0")
    ap.add_argument("--iters", type=int, default=300)
    args = ap.parse_args()
    from oracle import oracle as O

    for case in args.cases.split(","):
        kind, size = case.split(":")
        size = int(size)
        for sym in [int(v) for v in args.sym.split(",")]:
            for threads in [int(v) for v in args.threads.split(",")]:
                for parts in [int(v) for v in args.parts.split(",")]:
                    cfg = E.make_config(sym_pairs=sym, threads=threads, direct=2)
                    if kind == "fem3d":
                        nx, ny = FEM.get(size, (int(round((size / 3) ** (1 / 3))) + 1,) * 2)
                        m = E.Matrix.generate("fem3d", size, 3, nx, ny, 13500, 1, 1, cfg=cfg)
                    else:
                        m = E.Matrix.generate("kkt3d", size, cfg=cfg)
                    n = m.n
                    if parts:
                        m.c.nParts = parts
                        m.c.vectorCacheSize = min(65535, int(math.ceil(n / parts * 1.03)) + 2)
                    x = E.x_glibc(n)
                    y_ref = O.spmv_coo(n, m.I, m.J, m.V, x)
                    scale = O.abs_rowsum(n, m.I, m.J, m.V, x)
                    m.reorder(cfg)
                    plan = E.Plan(m, cfg)
                    st = plan.stats
                    perm = m.reorder_list.copy()
                    dx, dy = E.DeviceBuffer(n).upload(E.vector_reorder(x, perm)), E.DeviceBuffer(n)
                    r = plan.bench(dx.ptr, dy.ptr, warmup=20, iters=args.iters, per_kernel=False)
                    bad, worst = O.check_tolerance(E.vector_recover(dy.download(), perm), y_ref, scale)
                    us = r["ms_total"] / args.iters * 1e3
                    print(json.dumps({"case": case, "rows": n, "nnz": st["nnz"], "sym": sym, "threads": threads, "parts_asked": parts, "parts": st["n_parts"],
                                      "items": st["n_items"], "us": round(us, 2), "GFLOPs": round(2.0 * st["nnz"] / us / 1e3, 1),
                                      "frac_format": round(st["bytes_format"] / (us * 1e-6) / 8e12, 3), "fmt_MB": round(st["bytes_format"] / 1e6, 2),
                                      "halo_cols": st["halo_cols"], "nnz_er": st["nnz_er"], "lds_bytes": st["lds_bytes"], "rows_over_tol": bad}), flush=True)
                    plan.destroy()
                    m.free()


if __name__ == "__main__":
    main()
