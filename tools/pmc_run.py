#!/usr/bin/env python3
"""Workload for the PMC passes (run under `rocprofv3 --pmc <counter> --kernel-trace`):
a few launches of the streaming-read probe over a known byte count (calibration of FETCH_SIZE
for this access width, MI355X_MICROARCH.md section HBM) followed by SpMVs of the bench workload.

usage: rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d OUT -- python3 tools/pmc_run.py
"""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="audikw_1-like")
    ap.add_argument("--mtx", default="", help="a Matrix Market file instead of a generator workload (bench.py --mtx): storage from its banner as bench.py chooses it")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--plain", action="store_true", help="every entry stored even for a symmetric workload")
    ap.add_argument("--graph-compress", type=int, default=0, help="cfg.graph_compress (A/B of the partitions the k-way partitioner gives)")
    ap.add_argument("--layout-out", default="", help="write the plan's layout fingerprint (bench.layout_fingerprint) here: pmc_parse.py stores it with the entry")
    args = ap.parse_args()
    import bench as B
    import ehyb_spmv_gpu_amd as E
    from ehyb_spmv_gpu_amd import _lib

    lib = _lib.load()
    bw = C.c_double()
    lib.ehyb_measure_read_bw(1 << 30, 5, C.byref(bw))  # 8 launches of ehyb_read_kernel over 1 GiB
    if args.mtx:
        probe = E.Matrix.read_mtx(args.mtx)
        sym = probe.symmetric and probe.n >= B.SYM_MIN_ROWS and not args.plain   # as bench.py --mtx does
        probe.free()
        cfg = E.make_config(sym_pairs=1 if sym else 0, partitioner=B.partitioner_for(E, "file"))
        m = E.Matrix.read_mtx(args.mtx, cfg)
        args.workload = os.path.splitext(os.path.basename(args.mtx))[0]
    else:
        gen, gargs, _ = B.WORKLOADS[args.workload]
        sym = B.symmetric_storage_pays(gen, gargs) and not args.plain  # as bench.py does
        cfg = E.make_config(sym_pairs=1 if sym else 0, partitioner=B.partitioner_for(E, gen), graph_compress=args.graph_compress)
        m = E.Matrix.generate(gen, *gargs, cfg=cfg)
    x = E.x_glibc(m.n)
    m.reorder(cfg)
    xp = E.vector_reorder(x, m.reorder_list)
    plan = E.Plan(m, cfg)
    dx, dy = E.DeviceBuffer(m.n).upload(xp), E.DeviceBuffer(m.n)
    for _ in range(args.iters):
        plan.spmv(dx.ptr, dy.ptr)
    lib.ehyb_dev_sync()
    st = plan.stats
    if args.layout_out:
        import json

        json.dump(B.layout_fingerprint(st), open(args.layout_out, "w"))
    print("PMC_RUN", args.workload, "sym" if st["sym_pairs"] else "plain", {k: st[k] for k in ("nnz", "nnz_ell", "nnz_er", "size_block_ell", "bytes_format", "bytes_alg", "n_items", "window_loads")})


if __name__ == "__main__":
    main()
