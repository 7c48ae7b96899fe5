#!/bin/bash
# bench.py on every workload (one GPU box): one JSON line each into gpurun_out/<tag>_all_workloads.jsonl
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
TAG=${1:?tag}
OUT=gpurun_out/${TAG}_all_workloads.jsonl
: > $OUT
for W in audikw_1-like audikw_1-graded audikw_1-mesh banded-4M kkt3d-110 kkt3d-200 rmat-22 rmat-24 small bcsstk17-like; do
  timeout 900 python bench.py --workload $W --steps 100 --warmup 10 --no-cpu-baseline --no-dropin-arm --no-scaling-anchor 2> gpurun_out/${TAG}_$W.err | grep '^{' >> $OUT || echo "{\"workload\": \"$W\", \"error\": true}" >> $OUT
done
python - $OUT <<'PY'
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l)
    if "error" in d:
        print(d); continue
    r = d["roofline"]
    print(f'{d["config"]["workload"]:16s} {d["value"]:9.1f} GFLOP/s {d["ms_per_step"]*1e3:9.2f} us  {r["kernel"]:16s} frac {r["frac"]:.3f} alg_frac {r["alg_frac"]:.3f}  parity {d["parity"]["rows_over_1e-12"]}  plain {(d.get("plain_storage") or {}).get("value")}')
PY
