#!/bin/bash
# PMC traffic (FETCH_SIZE, WRITE_SIZE; separate passes) of every bench workload -> gpurun_out/pmc_traffic.json
# (copy to profiles/pmc_traffic.json: bench.py then quotes roofline.traffic for each of them)
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
export TMPDIR=/tmp
TAG=${1:?tag}
O=gpurun_out
cp profiles/pmc_traffic.json $O/pmc_traffic.json
# usage (inside gpurun): bash tools/pmc_all.sh <tag> ["workload ..."]
LIST=${2:-"audikw_1-graded banded-4M kkt3d-110 kkt3d-200 rmat-22 rmat-24 small bcsstk17-like"}
for W in $LIST; do
  STORAGE=$(python - "$W" <<'PY'
import sys
sys.path.insert(0, ".")
import bench as B
gen, gargs, _ = B.WORKLOADS[sys.argv[1]]
print("sym" if B.symmetric_storage_pays(gen, gargs) else "plain")
PY
)
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/${TAG}_${W}_pmc_fetch -- python3 tools/pmc_run.py --workload "$W" --iters 5 --layout-out $O/${TAG}_${W}_layout.json > $O/${TAG}_${W}_pmc_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/${TAG}_${W}_pmc_write -- python3 tools/pmc_run.py --workload "$W" --iters 5 > $O/${TAG}_${W}_pmc_write.log 2>&1
  python tools/pmc_parse.py $O/${TAG}_${W}_pmc_fetch $O/${TAG}_${W}_pmc_write $O/${TAG}_${W}_pmc_traffic.json --workload "$W" --storage "$STORAGE" --round "$TAG" --table $O/pmc_traffic.json --layout $O/${TAG}_${W}_layout.json > /dev/null
  rm -rf $O/${TAG}_${W}_pmc_fetch $O/${TAG}_${W}_pmc_write
  echo "$W $STORAGE done"
done
python - <<'PY'
import json
t = json.load(open("gpurun_out/pmc_traffic.json"))
for k, v in sorted(t["entries"].items()):
    print(f'{k:70s} {v["hbm_bytes_per_launch"] / 1e6:10.1f} MB  {v["evidence"]}')
PY
