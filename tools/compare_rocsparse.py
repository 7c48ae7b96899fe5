#!/usr/bin/env python3
"""EHYB next to the vendor library on the same matrix and GPU: rocSPARSE CSR SpMV (generic API,
several algorithms) vs the EHYB plan, both checked against the CPU oracle.  A comparison tool,
not part of the product path (the reference's cuSPARSE baselines, spmv.cu:135-437).

usage: python tools/compare_rocsparse.py [--workload audikw_1-like] [--iters 100]
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, "tools", "librocsparse_baseline.so")


def build():
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(os.path.join(ROOT, "tools", "rocsparse_baseline.cpp")):
        subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-shared", "-fPIC", "--offload-arch=gfx950",
                        os.path.join(ROOT, "tools", "rocsparse_baseline.cpp"), "-L/opt/rocm/lib", "-lrocsparse",
                        "-Wl,-rpath,/opt/rocm/lib", "-o", LIB], check=True)


def rocsparse_arm(E, O, np, gen, gargs, x, y_ref, scale, iters=100, algs=(("default", 0), ("adaptive", 2), ("lrb", 7))):
    """rocSPARSE CSR SpMV (generic API) on the same matrix, same GPU, same timing protocol; every
    result checked against the CPU oracle.  -> {"best": {...}, name: {...}} (bench.py --vendor-baseline)."""
    build()
    lib = C.CDLL(LIB)
    m = E.Matrix.read_mtx(gargs[0]) if gen == "file" else E.Matrix.generate(gen, *gargs)
    n, nnz = m.n, m.nnz
    rp, J, V = m.row_idx.copy(), m.J.copy(), m.V.copy()
    m.free()
    dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int)
    out = {}
    for name, alg in algs:
        y = np.zeros(n)
        ms, pre = C.c_double(), C.c_double()
        rc = lib.rocsparse_csr_spmv_bench(n, C.c_int64(nnz), rp.ctypes.data_as(ip), J.ctypes.data_as(ip),
                                          V.ctypes.data_as(dp), np.ascontiguousarray(x).ctypes.data_as(dp), y.ctypes.data_as(dp), alg, 10,
                                          iters, C.byref(ms), C.byref(pre))
        if rc != 0:
            out[name] = {"error": rc}
            continue
        bad, worst = O.check_tolerance(y, y_ref, scale)
        out[name] = {"ms_per_spmv": round(ms.value, 5), "GFLOP/s": round(2.0 * nnz / ms.value / 1e6, 2),
                     "preprocess_ms": round(pre.value, 3), "rows_over_1e-12": bad}
    ok = {k: v for k, v in out.items() if "ms_per_spmv" in v and v["rows_over_1e-12"] == 0}
    if ok:
        b = min(ok, key=lambda k: ok[k]["ms_per_spmv"])
        out["best"] = dict(ok[b], algorithm=b)
    out["library"] = "rocSPARSE rocsparse_spmv, CSR, fp64 (tools/rocsparse_baseline.cpp); not part of the product path"
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="audikw_1-like")
    ap.add_argument("--iters", type=int, default=100)
    ap.add_argument("--plain", action="store_true", help="no symmetric pair storage even for a symmetric workload")
    args = ap.parse_args()
    import numpy as np

    import bench as B
    import ehyb_spmv_gpu_amd as E
    from oracle import oracle as O

    build()
    lib = C.CDLL(LIB)
    gen, gargs, desc = B.WORKLOADS[args.workload]
    sym = B.symmetric_storage_pays(gen, gargs) and not args.plain   # as bench.py: symmetric pair storage for symmetric inputs
    cfg = E.make_config(sym_pairs=1 if sym else 0)
    m = E.Matrix.generate(gen, *gargs, cfg=cfg)
    n, nnz = m.n, m.nnz
    x = O.x_glibc(n)
    y_ref = O.spmv_coo(n, m.I, m.J, m.V, x)
    scale = O.abs_rowsum(n, m.I, m.J, m.V, x)
    rp, J, V = m.row_idx.copy(), m.J.copy(), m.V.copy()
    out = {"workload": args.workload, "rows": n, "nnz": nnz, "rocsparse": {}}
    dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int)
    for name, alg in (("default", 0), ("adaptive", 2), ("rowsplit", 3), ("lrb", 7), ("nnzsplit", 8)):
        y = np.zeros(n)
        ms, pre = C.c_double(), C.c_double()
        rc = lib.rocsparse_csr_spmv_bench(n, C.c_int64(nnz), rp.ctypes.data_as(ip), J.ctypes.data_as(ip),
                                          V.ctypes.data_as(dp), x.ctypes.data_as(dp), y.ctypes.data_as(dp), alg, 10,
                                          args.iters, C.byref(ms), C.byref(pre))
        if rc != 0:
            out["rocsparse"][name] = {"error": rc}
            continue
        bad, worst = O.check_tolerance(y, y_ref, scale)
        out["rocsparse"][name] = {"ms": round(ms.value, 5), "GFLOPs": round(2 * nnz / ms.value / 1e6, 1),
                                  "preprocess_ms": round(pre.value, 3), "rows_over_1e-12": bad, "worst": worst}
        print(f"rocsparse {name:9s}: {ms.value * 1e3:8.1f} us  {2 * nnz / ms.value / 1e6:8.1f} GFLOP/s  (preprocess {pre.value:.2f} ms, parity bad={bad})", flush=True)
    m.reorder(cfg)
    perm = m.reorder_list.copy()
    plan = E.Plan(m, cfg)
    dx, dy = E.DeviceBuffer(n).upload(E.vector_reorder(x, perm)), E.DeviceBuffer(n)
    r = plan.bench(dx.ptr, dy.ptr, warmup=10, iters=args.iters, per_kernel=False)
    ms = r["ms_total"] / args.iters
    bad, worst = O.check_tolerance(E.vector_recover(dy.download(), perm), y_ref, scale)
    out["ehyb"] = {"ms": round(ms, 5), "GFLOPs": round(2 * nnz / ms / 1e6, 1), "rows_over_1e-12": bad, "worst": worst,
                   "symmetric_pair_storage": bool(plan.stats["sym_pairs"])}
    print(f"ehyb{' (sym pairs)' if sym else '            '}   : {ms * 1e3:8.1f} us  {2 * nnz / ms / 1e6:8.1f} GFLOP/s  (parity bad={bad})")
    best = min(v["ms"] for v in out["rocsparse"].values() if "ms" in v)
    out["speedup_vs_best_rocsparse"] = round(best / ms, 3)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
