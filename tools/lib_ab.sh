#!/bin/bash
# A/B of several builds of the library on one box: bench lines of the R-MAT workloads, arms alternating.
#   usage (inside gpurun): bash tools/lib_ab.sh <tag> <other libehyb.so> [more ...]     e.g. _ab/libehyb_k12.so
# Arm "base" is the in-tree library; every other arm runs with EHYB_LIB set to the named file.
cd "${GRAFT_REPO_ROOT:?}"; export TMPDIR=/tmp; mkdir -p gpurun_out
TAG=${1:?tag}; shift
OUT=gpurun_out/${TAG}_lib_ab.jsonl; : > $OUT
for W in ${WORKLOADS:-rmat-22 rmat-24}; do
  for rep in 1 2; do
    for ARM in base "$@"; do
      if [ "$ARM" = base ]; then unset EHYB_LIB; else export EHYB_LIB=$PWD/$ARM; fi
      python bench.py --workload $W --steps 200 --warmup 20 --no-cpu-baseline --no-scaling-anchor --no-live-pmc 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(json.dumps({'arm':'$(basename $ARM)','workload':'$W','us':round(d['ms_per_step']*1e3,2),'gflops':d['value'],'parity':d['parity']['rows_over_1e-12']}))" | tee -a $OUT
    done
  done
done
