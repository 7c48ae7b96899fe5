#!/bin/bash
# A/B of two builds of the library on one box: bench lines of the R-MAT workloads, alternating.
#   usage (inside gpurun): bash tools/lib_ab.sh <tag> <path of the other libehyb.so, e.g. _ab/libehyb_x.so>   (arm "other" runs with EHYB_LIB set)
TAG=${1:?tag}; OTHER=${2:?other library}
cd "${GRAFT_REPO_ROOT:?}"; export TMPDIR=/tmp; mkdir -p gpurun_out
OUT=gpurun_out/${TAG}_lib_ab.jsonl; : > $OUT
for W in rmat-22 rmat-24; do
  for rep in 1 2; do
    for ARM in base other; do
      if [ $ARM = other ]; then export EHYB_LIB=$PWD/$OTHER; else unset EHYB_LIB; fi
      python bench.py --workload $W --steps 200 --warmup 20 --no-cpu-baseline --no-scaling-anchor 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(json.dumps({'arm':'$ARM','workload':'$W','us':round(d['ms_per_step']*1e3,2),'gflops':d['value'],'parity':d['parity']['rows_over_1e-12']}))" | tee -a $OUT
    done
  done
done
