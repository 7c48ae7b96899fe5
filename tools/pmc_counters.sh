#!/usr/bin/env bash
# Any PMC counters per kernel of the SpMV of one workload: rocprofv3 --pmc passes of at most four counters each (one process per
# pass, nothing but --kernel-trace beside --pmc), mean per launch.   usage (inside gpurun): bash tools/pmc_counters.sh <workload> CTR1 CTR2 ...
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
WL=${1:?workload}; shift
OUT=gpurun_out/pmc_ctr_$WL
rm -rf "$OUT"; mkdir -p "$OUT"
i=0
while [ $# -gt 0 ]; do
  GROUP=("${@:1:4}"); shift $(( $# < 4 ? $# : 4 ))
  rocprofv3 --pmc "${GROUP[@]}" --kernel-trace --output-format csv -d "$OUT/pass$i" -- python3 tools/pmc_run.py --workload "$WL" --iters 5 > "$OUT/pass$i.log" 2>&1 || echo "pass $i (${GROUP[*]}) failed"
  i=$((i+1))
done
python3 - "$OUT" <<'PY'
import csv, glob, collections, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in sorted(acc.items()):
    if "ehyb" not in k or "read_kernel" in k:
        continue
    print(k, {n: round(sum(v) / len(v)) for n, v in sorted(c.items())})
PY
