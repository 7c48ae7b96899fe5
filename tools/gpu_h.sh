#!/bin/bash
# round 2, GPU call H: reduce-pass structure sweep (row-block size x units), bare and full
cd "${GRAFT_REPO_ROOT:?}"
export TMPDIR=/tmp
O=gpurun_out/h
mkdir -p $O
for BR in 1024 2048 4096 8192; do for U2 in 1024 2048 4096; do for P in 0 255; do
  EHYB_PB_PROBE=$P EHYB_PB_UNITS2=$U2 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r -- python3 tools/er_ab.py --workloads rmat-22 --iters 20 --block-rows $BR > $O/r.log 2>&1
  python - $O/r $BR $U2 $P <<'PY'
import csv, glob, sys
out = {}
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "pb_" in r["Name"]:
            out["scale" if "scale" in r["Name"] else "reduce"] = round(float(r["AverageNs"]) / 1e3, 1)
print("block_rows", sys.argv[2], "units2", sys.argv[3], "probe", sys.argv[4], out)
PY
  rm -rf $O/r
done; done; done 2>&1 | tee $O/sweep.txt
