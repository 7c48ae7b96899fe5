#!/bin/bash
# round 2, GPU call F: evidence set r02_a (bench + rocprof stats + PMC), all workloads, rocSPARSE comparison, rmat-24 residual A/B
cd "${GRAFT_REPO_ROOT:?}"
export TMPDIR=/tmp
cp profiles/pmc_traffic.json gpurun_out/pmc_traffic.json
bash tools/evidence.sh r02_a audikw_1-like
bash tools/evidence.sh r02_a_plain audikw_1-like --sym-pairs off --no-dropin-arm --no-scaling-anchor
bash tools/all_workloads.sh r02_a
for W in audikw_1-like kkt3d-110 banded-4M rmat-22; do timeout 600 python tools/compare_rocsparse.py --workload $W --iters 50 2>/dev/null | tail -1; done > gpurun_out/r02_a_rocsparse.txt; cat gpurun_out/r02_a_rocsparse.txt | cut -c1-900
timeout 900 python tools/er_ab.py --workloads rmat-24 --iters 20 > gpurun_out/r02_a_er_ab_rmat24.jsonl 2>/dev/null; cat gpurun_out/r02_a_er_ab_rmat24.jsonl | cut -c1-400
timeout 600 python -m pytest tests/test_gpu_cli.py -m gpu -q 2>&1 | tail -3
