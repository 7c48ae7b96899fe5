#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}"
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/final_pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" gpurun_out/final_pytest.log | tail -2
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
cp profiles/pmc_traffic.json gpurun_out/pmc_traffic.json
bash tools/evidence.sh r02_c audikw_1-like
bash tools/all_workloads.sh r02_c
