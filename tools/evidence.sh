#!/bin/bash
# One evidence set on the GPU box: bench line, rocprofv3 kernel stats of the same command, PMC traffic.
#   usage (inside gpurun): bash tools/evidence.sh <tag> [workload] [extra bench args...]
# Writes gpurun_out/<tag>_*; copy what is to be judged into profiles/.
set -euo pipefail
cd "${GRAFT_REPO_ROOT:?}"
TAG=${1:?tag}
WL=${2:-audikw_1-like}
shift $(( $# > 1 ? 2 : 1 ))
EXTRA=("$@")
export TMPDIR=/tmp
O=gpurun_out
python bench.py --steps 200 --warmup 20 --workload "$WL" "${EXTRA[@]}" > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err || { tail -20 $O/${TAG}_bench.err; exit 1; }
tail -1 $O/${TAG}_bench.json | cut -c1-600
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof -- python3 bench.py --steps 200 --warmup 20 --workload "$WL" --no-cpu-baseline --no-plain-arm --no-dropin-arm --no-scaling-anchor --no-live-pmc "${EXTRA[@]}" > $O/${TAG}_bench_under_rocprof.json 2> $O/${TAG}_prof.err
cp "$(find $O/${TAG}_prof -name '*kernel_stats.csv' | head -1)" $O/${TAG}_kernel_stats.csv
head -5 $O/${TAG}_kernel_stats.csv
STORAGE=$(python - "$O/${TAG}_bench.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("sym" if d["config"]["sym_pairs"] else "plain")
PY
)
PLAIN=(); [ "$STORAGE" = plain ] && PLAIN=(--plain)
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/${TAG}_pmc_fetch -- python3 tools/pmc_run.py --workload "$WL" "${PLAIN[@]}" --layout-out $O/${TAG}_layout.json > $O/${TAG}_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/${TAG}_pmc_write -- python3 tools/pmc_run.py --workload "$WL" "${PLAIN[@]}" > $O/${TAG}_pmc_write.log 2>&1
python tools/pmc_parse.py $O/${TAG}_pmc_fetch $O/${TAG}_pmc_write $O/${TAG}_pmc_traffic.json --workload "$WL" --storage "$STORAGE" --round "$TAG" --table $O/pmc_traffic.json --layout $O/${TAG}_layout.json > /dev/null
grep -n "hbm_bytes_per_launch\|factor" $O/${TAG}_pmc_traffic.json | head
