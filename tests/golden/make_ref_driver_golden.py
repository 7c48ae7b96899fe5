#!/usr/bin/env python3
"""Golden output of the REFERENCE'S OWN driver (run on a GPU box; the fixtures are committed).

oracle/_ref/solver_test_ref is /root/reference/solver_test.c, unchanged, compiled against include/ and
linked with libehyb.so (oracle/Makefile).  Its main() reads ./read/<name>.mtx, accumulates the
reference CPU product y in file order (solver_test.c:102 / 247,254), calls matrixReorder ->
vectorReorder -> spmvGPuEHYB -> vectorRecover and prints ten rows of both vectors
("at %d yResult is %f y is  %f", solver_test.c:385-388) and its compare() line
("diff is %e, ampldiff is %e", solver_test.c:28).  The `y is` column is the output of the reference's
CPU path itself: tests/test_golden.py checks the oracle (oracle/ehyb_oracle.c, the restatement of that
path) against it without a GPU and without the reference tree.

usage (GPU box): python tests/golden/make_ref_driver_golden.py   -> gpurun_out/ref_driver_<tag>.txt (copy to tests/golden/)
The matrices come from tests/golden/ref_cases.py (deterministic generators + deterministic edits; tests/test_golden.py
rebuilds them the same way).
"""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ref_cases  # noqa: E402  (the matrices: tests/golden/ref_cases.py)


def main():
    import ehyb_spmv_gpu_amd as E

    exe = os.path.join(ROOT, "oracle", "_ref", "solver_test_ref")
    out_dir = os.path.join(ROOT, "gpurun_out") if os.path.isdir(os.path.join(ROOT, "gpurun_out")) else os.path.dirname(os.path.abspath(__file__))
    for tag in ref_cases.TAGS:
        with tempfile.TemporaryDirectory() as d:
            os.mkdir(os.path.join(d, "read"))
            m, sym = ref_cases.build(E, tag)
            m.write_mtx(os.path.join(d, "read", "a.mtx"), symmetric_lower_only=sym)
            p = subprocess.run([exe, "-m", "a", "-i", "20"], cwd=d, capture_output=True, text=True, timeout=600)
            assert p.returncode == 0, p.stdout + p.stderr
            # everything the driver itself prints about the product: its sizing, the ten rows, up to 100 rows its 1 % test
            # flags ("large difference at ...: realy <its CPU y> vs yResult <the GPU y>") and the two sums
            keep = [ln.strip() for ln in p.stdout.splitlines() if ln.strip().startswith(("at ", "diff is", "read ", "parts is", "maxCol", "large difference"))]
            open(os.path.join(out_dir, f"ref_driver_{tag}.txt"), "w").write("\n".join(keep) + "\n")
            print(tag, keep[-1])


if __name__ == "__main__":
    main()
