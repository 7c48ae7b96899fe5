#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}"
export TMPDIR=/tmp
O=gpurun_out
timeout 2400 python -m pytest tests -m gpu -q --durations=8 > $O/m_pytest.log 2>&1; echo "pytest rc=$?" >> $O/m_pytest.log
grep -E "^FAILED|^ERROR|passed|failed" $O/m_pytest.log | tail -12
bash tools/all_workloads.sh r02_b
timeout 600 python tools/er_ab.py --workloads rmat-22 --iters 30 --panel-cols 8192 --block-rows 2048 2>/dev/null | cut -c1-330
for W in rmat-22 rmat-24; do timeout 900 python tools/compare_rocsparse.py --workload $W --iters 30 2>/dev/null | tail -1; done > $O/r02_b_rocsparse_rmat.txt; cut -c1-1500 $O/r02_b_rocsparse_rmat.txt
