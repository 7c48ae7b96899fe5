#!/bin/bash
# round 2, GPU call A: box probe, full GPU test suite with durations, residual-kernel PMC on rmat-22, bench line
cd "${GRAFT_REPO_ROOT:?}"
export TMPDIR=/tmp
O=gpurun_out/a
mkdir -p $O
{ nproc; free -g; cat /sys/fs/cgroup/cpu.max 2>/dev/null; df -h /tmp | tail -1; } > $O/box.txt 2>&1
( find / -xdev \( -name "*.mtx" -o -name "*.mtx.gz" -o -name "audikw*" -o -name "nlpkkt*" -o -name "bcsstk*" \) -size +100k 2>/dev/null | head -20 ) > $O/mtx_probe.txt
ls /data /datasets ./read 2>&1 | head >> $O/mtx_probe.txt
timeout 1500 python -m pytest tests -m gpu -x -q --durations=25 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -40 $O/pytest.log
rocprofv3 -L > $O/counters.txt 2>&1
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_REQ_sum TCC_READ_sum"; do
  T=$(echo $C | tr ' ' '_')
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_$T -- python3 tools/pmc_run.py --workload rmat-22 --iters 5 > $O/pmc_$T.log 2>&1
  python - "$O/pmc_$T" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[(r["Kernel_Name"].split("(")[0][:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(k, len(v), sum(v) / len(v))
PY
done > $O/pmc_er_rmat22.txt 2>&1
cat $O/pmc_er_rmat22.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -3 $O/bench.err; cut -c1-3000 $O/bench.json
