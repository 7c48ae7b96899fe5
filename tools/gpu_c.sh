#!/bin/bash
# round 2, GPU call C: residual A/B after the fixes, failed tests again, small-matrix shape, bench line
cd "${GRAFT_REPO_ROOT:?}"
export TMPDIR=/tmp
O=gpurun_out/c
mkdir -p $O
timeout 600 python tools/er_ab.py --workloads rmat-22,kkt3d-110c --panel-cols 8192 --block-rows 8192 > $O/er_ab.jsonl 2> $O/er_ab.err; echo "er_ab rc=$?"; cat $O/er_ab.jsonl; tail -3 $O/er_ab.err
timeout 2400 python -m pytest tests -m gpu -q --durations=12 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
grep -E "^FAILED|^ERROR|passed|failed" $O/pytest.log | tail -30
for W in bcsstk17-like small; do for D in 1 2; do
python - $W $D <<'PY'
import sys, json
sys.path.insert(0, ".")
import bench as B, ehyb_spmv_gpu_amd as E
wl, d = sys.argv[1], int(sys.argv[2])
gen, gargs, _ = B.WORKLOADS[wl]
cfg = E.make_config(direct=d)
m = E.Matrix.generate(gen, *gargs, cfg=cfg); x = E.x_glibc(m.n); m.reorder(cfg)
plan = E.Plan(m, cfg)
dx, dy = E.DeviceBuffer(m.n).upload(E.vector_reorder(x, m.reorder_list)), E.DeviceBuffer(m.n)
r = plan.bench(dx.ptr, dy.ptr, warmup=50, iters=2000, per_kernel=False)
print(json.dumps({"workload": wl, "direct": d, "rows": m.n, "nnz": m.nnz, "us_spmv": round(r["ms_total"] / 2000 * 1e3, 3)}))
PY
done; done 2>&1 | tee $O/direct_ab.jsonl
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -3 $O/bench.err; cut -c1-5000 $O/bench.json
