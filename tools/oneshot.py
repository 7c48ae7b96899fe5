#!/usr/bin/env python3
"""What the one-shot drop-in call costs end to end: spmvGPuEHYB takes HOST
buffers, so one call = layout conversion + ~0.8 GB upload over PCIe + 10 warm-ups + MAXIter
multiplies + download, exactly the reference's sequence (spmv.cu:61-133).  bench.py's `value` is
the steady-state rate with inputs resident in HBM; this tool reports the inclusive figure beside it.

usage: python tools/oneshot.py [--workload audikw_1-like] [--iters 1,100,2000]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="audikw_1-like")
    ap.add_argument("--iters", default="1,100,2000")
    args = ap.parse_args()
    import bench as B
    import ehyb_spmv_gpu_amd as E

    gen, gargs, _ = B.WORKLOADS[args.workload]
    cfg = E.make_config()
    m = E.Matrix.generate(gen, *gargs, cfg=cfg)
    x = E.x_glibc(m.n)
    t0 = time.time()
    m.reorder(cfg)
    t_reorder = time.time() - t0
    xp = E.vector_reorder(x, m.reorder_list)
    t0 = time.time()
    plan = E.Plan(m, cfg, upload=False)
    t_layout = time.time() - t0
    t0 = time.time()
    plan.upload()
    E.DeviceBuffer(8).download()  # a synchronising call behind the uploads
    t_upload = time.time() - t0
    st = plan.stats
    up_bytes = 8 * (st["size_block_ell"] + st["er_inline"]) + 4 * st["col_words"] + 80 * st["n_slabs"] + 12 * st["nnz_er"]
    print(f"# {args.workload}: n={m.n} nnz={m.nnz}; reorder {t_reorder:.2f} s, layout {t_layout:.2f} s, "
          f"upload {t_upload * 1e3:.0f} ms for {up_bytes / 1e6:.0f} MB ({up_bytes / t_upload / 1e9:.1f} GB/s)")
    plan.destroy()
    E.spmv_gpu_ehyb(m, xp, 1)  # first call pays library/device initialisation
    for it in [int(v) for v in args.iters.split(",")]:
        t0 = time.time()
        E.spmv_gpu_ehyb(m, xp, it)
        dt = time.time() - t0
        print(f"spmvGPuEHYB MAXIter={it:5d}: {dt * 1e3:9.1f} ms wall for {it + 10} multiplies -> "
              f"{2.0 * m.nnz * it / dt / 1e9:8.2f} GFLOP/s inclusive of conversion, PCIe upload and download")


if __name__ == "__main__":
    main()
