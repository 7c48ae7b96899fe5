#!/bin/bash
# What the driver does at round end, plus the evidence sets: GPU test suite, smoke(), bench + rocprofv3 stats + PMC of the
# default workload (symmetric pairs and every entry stored) and of R-MAT 2^22, every workload.
#   usage (inside gpurun): bash tools/final_check.sh <tag>
cd "${GRAFT_REPO_ROOT:?}"
TAG=${1:-r03_d}
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/final_pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" gpurun_out/final_pytest.log | tail -2
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
cp profiles/pmc_traffic.json gpurun_out/pmc_traffic.json
bash tools/evidence.sh $TAG audikw_1-like
bash tools/evidence.sh ${TAG}_plain audikw_1-like --sym-pairs off
bash tools/evidence.sh ${TAG}_rmat22 rmat-22
cp gpurun_out/pmc_traffic.json profiles/pmc_traffic.json
bash tools/all_workloads.sh $TAG
