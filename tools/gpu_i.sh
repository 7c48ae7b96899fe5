#!/bin/bash
# round 2, GPU call I: panel residual with LDS piece sums: A/B on rmat-22 / rmat-24 / kkt contiguous, per-kernel split, parity tests
cd "${GRAFT_REPO_ROOT:?}"
export TMPDIR=/tmp
O=gpurun_out/i
mkdir -p $O
timeout 900 python tools/er_ab.py --workloads rmat-22,rmat-24,kkt3d-110c --iters 30 --panel-cols 4096,8192 --block-rows 2048 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d=json.loads(l); print(d['workload'], d['arm'], 'spmv', d['us_spmv'], 'ell', d['us_ell'], 'er', d['us_er'], 'bad', d['rows_over_tol'])
" | tee $O/er_ab.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 tools/er_ab.py --workloads rmat-22 --iters 30 --block-rows 2048 > $O/prof.log 2>&1
python - $O <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/prof/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        print(r["Name"][:44], r["Calls"], r["AverageNs"])
PY
timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_fuzz.py -m gpu -q 2>&1 | tail -4
