#!/usr/bin/env python3
"""Extended fuzz of the plan configurations on the GPU (tests/fuzz_cases.py: random matrices, every
residual form, direct shape, pruning, relative columns, symmetric pairs, windows from 64 doubles to
160 KiB): seeds beyond the 40 the test suite runs.  usage: python tools/fuzz_gpu.py [first] [count]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 400
    import ehyb_spmv_gpu_amd as E
    from fuzz_cases import build
    from oracle import oracle as O

    bad_seeds = []
    kinds = {"direct": 0, "panel": 0, "panel_built_on_device": 0, "inline": 0, "csr": 0, "sym": 0, "empty_residual": 0}
    for seed in range(first, first + count):
        m, cfg, kw, x, y_ref, scale = build(E, O, seed)
        cfg.value_map = 1   # slot maps, for the refill below
        plan = E.Plan(m, cfg)
        st = plan.stats
        xp = E.vector_reorder(x, m.reorder_list)
        y = E.vector_recover(plan.spmv_host(xp, iters=2), m.reorder_list)
        bad, worst = O.check_tolerance(y, y_ref, scale)
        # numeric phase on the device: A -> -2 A on the same pattern (exact), the multiply again
        plan.set_values(-2.0 * m.V)
        bad3, worst3 = O.check_tolerance(E.vector_recover(plan.spmv_host(xp), m.reorder_list), -2.0 * y_ref, 2.0 * scale)
        bad, worst = bad + bad3, max(worst, worst3)
        direct = st["nnz_ell"] == 0 and st["nnz_er"] == st["nnz"] and st["er_segments"] == m.n
        kinds["direct"] += direct
        kinds["panel"] += st["er_partials"] > 0
        kinds["panel_built_on_device"] += st["er_partials"] > 0 and st["er_segments"] == 0
        kinds["inline"] += st["er_inline"] > 0
        kinds["csr"] += (not direct) and st["nnz_er"] > 0 and st["er_partials"] == 0 and st["er_inline"] == 0
        kinds["sym"] += st["sym_pairs"] > 0
        kinds["empty_residual"] += st["nnz_er"] == 0
        if bad:
            bad_seeds.append((seed, kw, bad, worst))
            print("FAIL", seed, kw, bad, worst, flush=True)
        plan.destroy()
    print(f"fuzz_gpu: seeds {first}..{first + count - 1}: {len(bad_seeds)} failures; plan kinds {kinds}")
    sys.exit(1 if bad_seeds else 0)


if __name__ == "__main__":
    main()
