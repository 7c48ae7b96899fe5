#!/bin/bash
# round 2, GPU call E: tuned panel residual, graded surrogate with entry-balanced partitions, relative columns on the band, suite
cd "${GRAFT_REPO_ROOT:?}"
export TMPDIR=/tmp
O=gpurun_out/e
mkdir -p $O
for U in "2048 2048" "4096 2048" "1024 1024"; do set -- $U
  EHYB_PB_UNITS1=$1 EHYB_PB_UNITS2=$2 timeout 300 python tools/er_ab.py --workloads rmat-22 --iters 50 --panel-cols 4096,8192 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d=json.loads(l); print('units', '$1', '$2', d['arm'], 'spmv', d['us_spmv'], 'ell', d['us_ell'], 'er', d['us_er'], 'bad', d['rows_over_tol'])
"
done 2>&1 | tee $O/pb_units.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_pb -- python3 tools/er_ab.py --workloads rmat-22 --iters 50 > $O/prof_pb.log 2>&1
python - $O <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/prof_pb/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        print(r["Name"][:44], r["Calls"], r["AverageNs"])
PY
for W in audikw_1-graded banded-4M kkt3d-110; do
timeout 600 python bench.py --workload $W --no-cpu-baseline --no-dropin-arm --no-scaling-anchor > $O/bench_$W.json 2> $O/bench_$W.err; echo "$W rc=$?"; tail -1 $O/bench_$W.err
python - $O/bench_$W.json <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d = json.loads(l); r = d["roofline"]
        print(d["config"]["workload"], d["value"], "GFLOP/s", d["ms_per_step"], "ms frac", r["frac"], "alg_frac", r["alg_frac"], "parts", d["config"]["parts"], "plain", (d.get("plain_storage") or {}).get("value"))
PY
done
timeout 2400 python -m pytest tests -m gpu -q --durations=8 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
grep -E "^FAILED|^ERROR|passed|failed" $O/pytest.log | tail -20
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -2 $O/bench.err; cut -c1-700 $O/bench.json
