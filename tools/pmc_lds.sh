#!/usr/bin/env bash
# LDS pipe occupancy of the SpMV kernels: SQ_LDS_BANK_CONFLICT (extra cycles) over SQ_LDS_IDX_ACTIVE
# (all LDS-array cycles), plus wave cycles, one rocprofv3 --pmc pass (MI355X_MICROARCH.md, LDS section).
set -euo pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
OUT=gpurun_out/pmc_lds
rm -rf "$OUT"
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY --kernel-trace --output-format csv -d "$OUT" -- python3 tools/pmc_run.py --iters 5 "$@" > "$OUT.log" 2>&1   # e.g. --plain, --workload kkt3d-110
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_lds/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in acc.items():
    if "ehyb" not in k:
        continue
    m = {n: sum(v) / len(v) for n, v in c.items()}
    print(k, {n: round(v) for n, v in m.items()},
          "conflict/active = %.2f" % (m.get("SQ_LDS_BANK_CONFLICT", 0) / max(1, m.get("SQ_LDS_IDX_ACTIVE", 1))),
          "wait/wave = %.2f" % (m.get("SQ_WAIT_ANY", 0) / max(1, m.get("SQ_WAVE_CYCLES", 1))))
PY
