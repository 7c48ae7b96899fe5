#!/bin/bash
cd "${GRAFT_REPO_ROOT:?}"
export TMPDIR=/tmp
O=gpurun_out/o
mkdir -p $O
export EHYB_BENCH_ONE_DEVICE=1 EHYB_BENCH_BACKEND=gloo
for A in "4 halo rmat-22" "3 allgather rmat-22" "4 halo audikw_1-like"; do set -- $A
  timeout 900 python bench.py --gpus $1 --exchange $2 --workload $3 --steps 10 --warmup 2 > $O/strong_$1_$2_$3.json 2> $O/strong_$1_$2_$3.err; echo "rc=$? $A"
done
unset EHYB_BENCH_ONE_DEVICE EHYB_BENCH_BACKEND
python - <<'PY'
import sys, time
sys.path.insert(0, ".")
import ehyb_spmv_gpu_amd as E
cfg = E.make_config(sym_pairs=1, verbose=1)
m = E.Matrix.generate("fem3d", 943695, 3, 68, 68, 13500, 1, 1, cfg=cfg)
t0 = time.time(); m.reorder(cfg); print("reorder on this box: %.2fs with %d host threads" % (time.time() - t0, E.host_threads()))
PY
