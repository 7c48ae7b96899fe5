#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}"
export TMPDIR=/tmp
cp profiles/pmc_traffic.json gpurun_out/pmc_traffic.json
bash tools/evidence.sh r02_c_rmat22 rmat-22 --no-scaling-anchor
python - <<'PY'
import json
d = json.load(open("gpurun_out/r02_c_rmat22_pmc_traffic.json"))
for k, v in d["kernels"].items():
    print(k[:60], v["launches"], v["hbm_bytes_per_launch"])
PY
timeout 600 python bench.py --workload rmat-22 --no-cpu-baseline --no-scaling-anchor 2>/dev/null | cut -c1-1600
