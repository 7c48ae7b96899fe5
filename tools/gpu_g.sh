#!/bin/bash
# round 2, GPU call G: where the time of the two panel passes goes (probe variants under rocprofv3)
cd "${GRAFT_REPO_ROOT:?}"
export TMPDIR=/tmp
O=gpurun_out/g
mkdir -p $O
for P in 0 1 2 3 4 8 16 32 64 128 96 255; do
  EHYB_PB_PROBE=$P rocprofv3 --kernel-trace --stats --output-format csv -d $O/p$P -- python3 tools/er_ab.py --workloads rmat-22 --iters 30 > $O/p$P.log 2>&1
  python - $O/p$P $P <<'PY'
import csv, glob, sys
out = {}
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "pb_" in r["Name"]:
            out["scale" if "scale" in r["Name"] else "reduce"] = round(float(r["AverageNs"]) / 1e3, 1)
print("probe", sys.argv[2], out)
PY
done 2>&1 | tee $O/probe.txt
