#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}"
export TMPDIR=/tmp
O=gpurun_out/l
mkdir -p $O
for PCT in 110 90 70 50 1; do
EHYB_PRUNE_PCT=$PCT timeout 900 python tools/er_ab.py --workloads rmat-22,rmat-24 --iters 30 --panel-cols 8192 --block-rows 2048 2>/dev/null | grep -v "windows-kept\|\"csr\"" | python -c "
import sys, json
for l in sys.stdin:
    d=json.loads(l); print('pct', $PCT, d['workload'], d['arm'], 'spmv', d['us_spmv'], 'ell', d['us_ell'], 'er', d['us_er'], 'nnz_ell', d['nnz_ell'], 'bad', d['rows_over_tol'])
"
done | tee $O/prune.txt
