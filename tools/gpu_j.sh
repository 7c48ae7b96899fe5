#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}"
export TMPDIR=/tmp
O=gpurun_out/j
mkdir -p $O
for P in 0 1 2 3 256 8; do
  EHYB_PB_PROBE=$P rocprofv3 --kernel-trace --stats --output-format csv -d $O/r -- python3 tools/er_ab.py --workloads rmat-22 --iters 20 --block-rows 2048 > $O/r.log 2>&1
  python - $O/r $P <<'PY'
import csv, glob, sys
out = {}
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "pb_" in r["Name"]:
            out["scale" if "scale" in r["Name"] else "reduce"] = round(float(r["AverageNs"]) / 1e3, 1)
print("probe", sys.argv[2], out)
PY
  rm -rf $O/r
done 2>&1 | tee $O/probe.txt
for BR in 2048 8192 16384; do for U2 in 256 512 1024; do
  EHYB_PB_UNITS2=$U2 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r -- python3 tools/er_ab.py --workloads rmat-22 --iters 20 --block-rows $BR > $O/r.log 2>&1
  python - $O/r $BR $U2 <<'PY'
import csv, glob, sys
out = {}
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "pb_" in r["Name"]:
            out["scale" if "scale" in r["Name"] else "reduce"] = round(float(r["AverageNs"]) / 1e3, 1)
print("block_rows", sys.argv[2], "units2", sys.argv[3], out)
PY
  rm -rf $O/r
done; done 2>&1 | tee -a $O/probe.txt
timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_fuzz.py -m gpu -q > $O/pytest.log 2>&1; grep -E "passed|failed|^FAILED" $O/pytest.log | tail -5
