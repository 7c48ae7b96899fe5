#!/usr/bin/env python3
"""Medium-size fuzz on the GPU: the generators at 20 k - 2 M rows with random plan configurations (window
size, threads, residual form and its panel/block sizes, symmetric pairs, direct shape, pruning, column
sharing, partitioner), every result against the CPU oracle, plus the two-phase call and a second x.
usage: python tools/fuzz_gpu_medium.py [first_seed] [count]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def pick(rng):
    kind = rng.choice(["fem3d", "fem3d_graded", "rmat", "kkt3d", "stencil2d", "banded", "mesh3d"])
    if kind == "mesh3d":
        nodes = int(rng.integers(8000, 200000))
        return kind, (nodes * 3, 3, int(rng.integers(6, 26)), int(rng.choice([0, 1500, 2500])), int(rng.integers(1, 99)))
    if kind == "fem3d":
        nodes = int(rng.integers(8000, 250000))
        nx = int(rng.integers(12, 60))
        return kind, (nodes * 3, 3, nx, int(rng.integers(12, 60)), int(rng.choice([0, 13500, 200000])), int(rng.integers(0, 2)), int(rng.integers(1, 99)))
    if kind == "fem3d_graded":
        nodes = int(rng.integers(8000, 150000))
        return kind, (nodes * 3, 3, int(rng.integers(12, 50)), int(rng.integers(12, 50)), int(rng.choice([50000, 100000, 300000])),
                      int(rng.choice([300000, 705000])), int(rng.integers(0, 2)), int(rng.integers(1, 99)))
    if kind == "rmat":
        scale = int(rng.integers(15, 22))
        return kind, (scale, int((1 << scale) * rng.choice([2, 8, 16])), int(rng.integers(1, 99)))
    if kind == "kkt3d":
        return kind, (int(rng.integers(20, 90)),)
    if kind == "stencil2d":
        nx = int(rng.integers(150, 1200))
        return kind, (nx, int(rng.integers(150, 1200)), int(rng.choice([5, 9])), int(rng.choice([0, 1000, 50000])), int(rng.integers(1, 99)))
    n = 1024 * int(rng.integers(20, 2000))
    return kind, (n, int(rng.choice([8, 32, 64])), 1024)


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    import ehyb_spmv_gpu_amd as E
    from oracle import oracle as O

    fails = 0
    for seed in range(first, first + count):
        rng = np.random.default_rng(seed)
        kind, args = pick(rng)
        sym_ok = kind in ("fem3d", "fem3d_graded", "kkt3d", "stencil2d", "mesh3d")
        kw = dict(lds_doubles=int(rng.choice([0, 0, 4096, 10240, 20480])), threads=int(rng.choice([0, 256, 512, 1024])),
                  er_mode=int(rng.choice([0, 1, 2])), er_panel_cols=int(rng.choice([0, 1024, 4096, 16384])),
                  er_block_rows=int(rng.choice([0, 512, 8192])), direct=int(rng.choice([0, 0, 2])), ell_prune=int(rng.choice([1, 2])),
                  col_sharing=int(rng.choice([1, 2])), hub_rule=int(rng.choice([1, 2])), fuse_er=int(rng.choice([0, 0, 2])),
                  partitioner=int(rng.choice([0, 0, 1, 4])), sym_pairs=int(rng.integers(0, 2)) if sym_ok or rng.random() < 0.2 else 0,
                  window_mode=int(rng.choice([0, 0, 0, 1])),
                  er_sums=int(rng.choice([0, 0, 2])), er_queue=int(rng.choice([0, 1])), xcd_map=int(rng.choice([0, 2])),
                  er_panel_threads=int(rng.choice([0, 512, 1024])), er_units1=int(rng.choice([0, 0, 100, 5000])), graph_compress=int(rng.choice([0, 1, 2])),
                  symbolic=int(rng.choice([0, 1])))   # 0: a certain panel form is built on the device, 1: on the host
        if kw["window_mode"] == 1:
            kw["sym_pairs"] = 0
        kw = {k: v for k, v in kw.items() if v}
        t0 = time.time()
        cfg = E.make_config(**kw)
        m = E.Matrix.generate(kind, *args, cfg=cfg)
        n = m.n
        x = O.x_glibc(n)
        y_ref = O.spmv_coo(n, m.I, m.J, m.V, x)
        scale = O.abs_rowsum(n, m.I, m.J, m.V, x)
        m.reorder(cfg)
        perm = m.reorder_list.copy()
        cfg.value_map = 1   # slot maps, for the refill at the end
        plan = E.Plan(m, cfg)
        st = plan.stats
        y = E.vector_recover(plan.spmv_host(E.vector_reorder(x, perm), iters=2), perm)
        bad, worst = O.check_tolerance(y, y_ref, scale)
        # a second x through the two-phase call (where the plan has phases)
        bad2 = 0
        if not (st["nnz_ell"] == 0 and st["er_segments"] == n):
            x2 = x[::-1].copy()
            dx, dy = E.DeviceBuffer(n).upload(E.vector_reorder(x2, perm)), E.DeviceBuffer(n)
            plan.spmv(dx.ptr, dy.ptr, phase=1)
            plan.spmv(dx.ptr, dy.ptr, phase=2)
            y2 = E.vector_recover(dy.download(), perm)
            mI, mJ, mV = m.I, m.J, m.V   # permuted matrix: check in the permuted numbering
            ref2 = O.spmv_coo(n, mI, mJ, mV, E.vector_reorder(x2, perm))
            sc2 = O.abs_rowsum(n, mI, mJ, mV, E.vector_reorder(x2, perm))
            bad2, _ = O.check_tolerance(dy.download(), ref2, sc2)
            dx.free(), dy.free()
        # numeric phase on the device: A -> -2 A on the same pattern (exact), the multiply again
        plan.set_values(-2.0 * m.V)
        y3 = E.vector_recover(plan.spmv_host(E.vector_reorder(x, perm)), perm)
        bad3, _ = O.check_tolerance(y3, -2.0 * y_ref, 2.0 * scale)
        bad2 += bad3
        form = ("direct" if st["nnz_ell"] == 0 and st["er_segments"] == n else "panel" if st["er_partials"] else
                "inline" if st["er_inline"] else "csr" if st["nnz_er"] else "pure-ell") + ("+sym" if st["sym_pairs"] else "") + (
                    "@dev" if st["er_partials"] and st["er_segments"] == 0 else "")
        print(f"seed {seed:4d} {kind:12s} n={n:8d} nnz={m.nnz:10d} {form:12s} bad={bad} bad2={bad2} worst={worst:.1e} {time.time() - t0:5.1f}s {kw}", flush=True)
        fails += (bad > 0) + (bad2 > 0)
        plan.destroy()
        m.free()
    print(f"fuzz_gpu_medium: {count} cases, {fails} failures")
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()
