#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}"
export TMPDIR=/tmp
O=gpurun_out/k
mkdir -p $O
timeout 900 python tools/er_ab.py --workloads rmat-22,rmat-24 --iters 30 --panel-cols 8192 --block-rows 2048 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d=json.loads(l); print(d['workload'], d['arm'], 'spmv', d['us_spmv'], 'ell', d['us_ell'], 'er', d['us_er'], 'nnz_ell', d['nnz_ell'], 'bad', d['rows_over_tol'])
" | tee $O/er_ab.txt
timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_fuzz.py -m gpu -q > $O/pytest.log 2>&1; grep -E "passed|failed|^FAILED" $O/pytest.log | tail -5
