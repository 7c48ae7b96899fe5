#!/bin/bash
# round 2, GPU call D: panel residual tuning (units per pass), per-kernel split under rocprofv3, graded surrogate
cd "${GRAFT_REPO_ROOT:?}"
export TMPDIR=/tmp
O=gpurun_out/d
mkdir -p $O
for U in "2048 2048" "1024 1024" "768 512" "512 512" "1024 256"; do set -- $U
  EHYB_PB_UNITS1=$1 EHYB_PB_UNITS2=$2 timeout 300 python tools/er_ab.py --workloads rmat-22 --iters 50 2>/dev/null | grep panel | python -c "
import sys, json
for l in sys.stdin:
    d=json.loads(l); print('units', '$1', '$2', d['arm'], 'spmv', d['us_spmv'], 'er', d['us_er'], 'bad', d['rows_over_tol'])
"
done 2>&1 | tee $O/pb_units.txt
EHYB_PB_UNITS1=1024 EHYB_PB_UNITS2=1024 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_pb -- python3 tools/er_ab.py --workloads rmat-22 --iters 50 > $O/prof_pb.log 2>&1
cat $(find $O/prof_pb -name '*kernel_stats.csv' | head -1) | cut -d, -f1-8 | head -12
timeout 600 python bench.py --workload audikw_1-graded --no-cpu-baseline --no-dropin-arm --no-scaling-anchor > $O/bench_graded.json 2> $O/bench_graded.err; echo "graded rc=$?"; tail -2 $O/bench_graded.err; cut -c1-1800 $O/bench_graded.json
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_fuzz.py tests/test_gpu_parity.py -m gpu -q 2>&1 | tail -5
