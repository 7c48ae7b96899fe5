#!/usr/bin/env python3
"""What new VALUES on an unchanged pattern cost: the whole pre-step again
(ehyb_matrix_reorder + layout + upload, as the reference repeats mt-metis + COO2EHYB + upload for every
matrix: solver_test.c:369-382, spmv.cu:74-99) against the numeric phase alone on the device
(ehyb_plan_set_values, csrc/ehyb_fill.hip) -- from a host array (PCIe upload included) and from a
device-resident one.  The refilled plan is checked against the CPU oracle on the new values.

usage: python tools/refill_time.py [--workload audikw_1-like] [--sym 0|1]
"""
import argparse
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="audikw_1-like")
    ap.add_argument("--sym", type=int, default=1)
    args = ap.parse_args()
    import numpy as np

    import bench as B
    import ehyb_spmv_gpu_amd as E
    from oracle import oracle as O

    gen, gargs, _ = B.WORKLOADS[args.workload]
    sym = bool(args.sym) and B.symmetric_storage_pays(gen, gargs)
    cfg = E.make_config(value_map=1, partitioner=B.partitioner_for(E, gen), **({"sym_pairs": 1} if sym else {}))
    m = E.Matrix.generate(gen, *gargs, cfg=cfg)
    n, nnz = m.n, m.nnz
    I0, J0, rp0 = m.I.copy(), m.J.copy(), m.row_idx.copy()
    x = E.x_glibc(n)
    t0 = time.time()
    m.reorder(cfg)
    t_reorder = time.time() - t0
    t0 = time.time()
    plan = E.Plan(m, cfg, upload=False)
    t_layout = time.time() - t0
    t0 = time.time()
    plan.upload()
    plan.lib.ehyb_dev_sync()
    t_upload = time.time() - t0
    st = plan.stats
    perm = m.reorder_list.copy()
    xp = E.vector_reorder(x, perm)
    order = E.entry_order(rp0, perm)
    print(f"# {args.workload} ({'symmetric pairs' if st['sym_pairs'] else 'every entry stored'}): n={n} nnz={nnz} "
          f"stored values {st['size_block_ell']} + residual {st['nnz_er']}")
    print(f"whole pre-step (what a new matrix costs without the numeric phase): reorder {t_reorder:.2f} s + layout {t_layout:.2f} s "
          f"+ upload {t_upload:.2f} s = {t_reorder + t_layout + t_upload:.2f} s")

    def vals(k):  # symmetric in (i, j), different for every k
        a, b = np.minimum(I0, J0).astype(np.int64), np.maximum(I0, J0).astype(np.int64)
        return ((a * 2654435761 + b * 40503 + 977 * k) % 2003 - 1001).astype(np.float64) / 977.0 + 0.0005

    lib = plan.lib
    V1 = vals(1)
    t0 = time.time()
    plan.set_values(V1, entry_order=order)  # first call also uploads the slot maps
    t_first = time.time() - t0
    V2 = vals(2)
    t0 = time.time()
    plan.set_values(V2, entry_order=order)
    t_host = time.time() - t0
    print(f"ehyb_plan_set_values from HOST arrays in the caller's original order (values + entry order over PCIe, "
          f"{(V2.nbytes + order.nbytes) / 1e6:.0f} MB): first call {t_first * 1e3:.1f} ms (slot maps uploaded), then {t_host * 1e3:.1f} ms")
    dv, do = C.c_void_p(), C.c_void_p()
    assert lib.ehyb_dev_alloc(V2.nbytes, C.byref(dv)) == 0 and lib.ehyb_dev_alloc(order.nbytes, C.byref(do)) == 0
    lib.ehyb_h2d(do, order.ctypes.data_as(C.c_void_p), order.nbytes)
    V3 = vals(3)
    lib.ehyb_h2d(dv, V3.ctypes.data_as(C.c_void_p), V3.nbytes)
    lib.ehyb_dev_sync()
    best = 1e9
    for _ in range(5):
        t0 = time.time()
        plan.set_values((dv.value, nnz), entry_order=(do.value, nnz))
        lib.ehyb_dev_sync()
        best = min(best, time.time() - t0)
    moved = 4 * (st["size_block_ell"] + st["er_inline"]) * (2 if st["sym_pairs"] else 1) + 8 * nnz + 4 * nnz + 8 * (st["size_block_ell"] + st["er_inline"])
    print(f"ehyb_plan_set_values from DEVICE-resident values + entry order: {best * 1e3:.3f} ms (best of 5; check + fill kernels, "
          f"~{moved / 1e6:.0f} MB moved = {moved / best / 1e9:.0f} GB/s)")
    y = E.vector_recover(plan.spmv_host(xp), perm)
    bad, worst = O.check_tolerance(y, O.spmv_coo(n, I0, J0, V3, x), O.abs_rowsum(n, I0, J0, V3, x))
    print(f"parity of the refilled plan against the CPU oracle on the new values: {bad} rows over 1e-12, worst {worst:.3e}")
    print(f"speed-up over repeating the pre-step: {(t_reorder + t_layout + t_upload) / t_host:.0f}x (host values), "
          f"{(t_reorder + t_layout + t_upload) / best:.0f}x (device values)")
    if bad:
        raise SystemExit(1)


if __name__ == "__main__":
    main()
