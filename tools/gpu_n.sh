#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}"
export TMPDIR=/tmp
O=gpurun_out/n
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_sym.py tests/test_gpu_fuzz.py tests/test_gpu_parity.py tests/test_gpu_cg.py -m gpu -q 2>&1 | tail -3
for i in 1 2; do
EHYB_LIB=$PWD/_ab/libehyb_base.so timeout 300 python bench.py --no-cpu-baseline --no-plain-arm --no-dropin-arm --no-scaling-anchor 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('BASE', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])
"
timeout 300 python bench.py --no-cpu-baseline --no-plain-arm --no-dropin-arm --no-scaling-anchor 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('PREFETCH', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])
"
done
for W in kkt3d-110 audikw_1-graded small; do
EHYB_LIB=$PWD/_ab/libehyb_base.so timeout 300 python bench.py --workload $W --no-cpu-baseline --no-plain-arm --no-dropin-arm --no-scaling-anchor 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('BASE', '$W', d['value'], d['ms_per_step'])
"
timeout 300 python bench.py --workload $W --no-cpu-baseline --no-plain-arm --no-dropin-arm --no-scaling-anchor 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('PREFETCH', '$W', d['value'], d['ms_per_step'])
"
done
