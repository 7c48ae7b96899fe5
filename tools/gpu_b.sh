#!/bin/bash
# round 2, GPU call B: whole GPU suite, residual A/B (CSR vs panel form), bench line
cd "${GRAFT_REPO_ROOT:?}"
export TMPDIR=/tmp
O=gpurun_out/b
mkdir -p $O
timeout 600 python tools/er_ab.py --workloads rmat-22,kkt3d-110c --panel-cols 8192,16384 --block-rows 8192 > $O/er_ab.jsonl 2> $O/er_ab.err; echo "er_ab rc=$?"; cat $O/er_ab.jsonl; tail -3 $O/er_ab.err
timeout 2400 python -m pytest tests -m gpu -q --durations=25 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -45 $O/pytest.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -3 $O/bench.err; cut -c1-4000 $O/bench.json
