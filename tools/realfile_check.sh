#!/bin/bash
# The headline matrix as a FILE at full size (round-3 verdict item 5): the audikw_1-like matrix written once as a symmetric
# Matrix Market file (plain and gzip, ~39 M stored lines, 1.2 GB of text) on the GPU box, then read back by the two programs a user
# of the reference would run: `solver_test -m <name> -i <iters>` (./read/<name>.mtx, as README.md:10 of the reference) and
# `bench.py --mtx`.  Prints the time-to-first-SpMV pieces: reader, reorder, plan, first multiply, plan-cache hit.
#   usage (inside gpurun): bash tools/realfile_check.sh <tag>
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
TAG=${1:?tag}
O=$GRAFT_REPO_ROOT/gpurun_out
W=/tmp/ehyb_realfile
mkdir -p $W/read $O
python - "$W" > $O/${TAG}_realfile_write.txt 2>&1 <<'PY'
import os, sys, time
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import ehyb_spmv_gpu_amd as E
w = sys.argv[1]
t = time.time(); m = E.Matrix.generate("fem3d", 943695, 3, 68, 68, 13500, 1, 1); print(f"generated: {m.n} rows, {m.nnz} entries in {time.time() - t:.2f} s")
t = time.time(); m.write_mtx(f"{w}/read/audikw_1_like.mtx", True); dt = time.time() - t
sz = os.path.getsize(f"{w}/read/audikw_1_like.mtx")
print(f"written: {sz / 1e6:.1f} MB of text in {dt:.2f} s ({sz / 1e6 / dt:.0f} MB/s)")
for rep in range(2):
    t = time.time(); r = E.Matrix.read_mtx(f"{w}/read/audikw_1_like.mtx", E.make_config(verbose=1)); dt = time.time() - t
    print(f"ehyb_mm_read (plain, pass {rep}): {dt:.2f} s = {sz / 1e6 / dt:.0f} MB/s of text; {r.n} rows, {r.nnz} entries expanded, symmetric={r.symmetric}")
    r.free()
PY
cat $O/${TAG}_realfile_write.txt
( cd $W/read && T0=$(date +%s.%N) && gzip -1 -k -f audikw_1_like.mtx && mv audikw_1_like.mtx.gz audikw_1_like_gz.mtx.gz && echo "gzip -1: $(python3 -c "import sys,time; print(round(time.time() - float(sys.argv[1]), 2))" $T0) s" && ls -la ) 2>&1 | tail -4
python - "$W" >> $O/${TAG}_realfile_write.txt 2>&1 <<'PY'
import os, sys, time
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import ehyb_spmv_gpu_amd as E
w = sys.argv[1]
sz = os.path.getsize(f"{w}/read/audikw_1_like.mtx")
t = time.time(); r = E.Matrix.read_mtx(f"{w}/read/audikw_1_like_gz.mtx", E.make_config(verbose=1)); dt = time.time() - t   # finds <name>.mtx.gz
print(f"ehyb_mm_read (gzip): {dt:.2f} s = {sz / 1e6 / dt:.0f} MB/s of text ({os.path.getsize(w + '/read/audikw_1_like_gz.mtx.gz') / 1e6:.0f} MB compressed); {r.n} rows, {r.nnz} entries")
PY
tail -5 $O/${TAG}_realfile_write.txt
# the reference's way: ./read/<name>.mtx relative to the working directory, -m name -i iters (solver_test.c:284,318-321); -c = the plan cache
cd $W
for pass in miss hit; do
  T0=$(date +%s.%N)
  $GRAFT_REPO_ROOT/ehyb_spmv_gpu_amd/solver_test -m audikw_1_like -i 2000 -c $W/plan.cache -v > $O/${TAG}_solver_test_$pass.txt 2>&1
  echo "solver_test wall clock (plan cache $pass): $(python3 -c "import sys,time; print(round(time.time() - float(sys.argv[1]), 2))" $T0) s" | tee -a $O/${TAG}_solver_test_$pass.txt
  grep -a "filename\|parts is\|reorder time\|plan cache\|iter is\|strict check\|PASSED\|FAILED\|wall clock\|symmetric pair\|CPU reference" $O/${TAG}_solver_test_$pass.txt
done
cd $GRAFT_REPO_ROOT
python bench.py --mtx $W/read/audikw_1_like.mtx --steps 200 --warmup 20 --no-scaling-anchor > $O/${TAG}_bench_mtx.json 2> $O/${TAG}_bench_mtx.err || tail -5 $O/${TAG}_bench_mtx.err
grep -a "^\[bench\]" $O/${TAG}_bench_mtx.err | head -12
python bench.py --steps 200 --warmup 20 --no-scaling-anchor --no-dropin-arm --no-plain-arm > $O/${TAG}_bench_generator.json 2> $O/${TAG}_bench_generator.err || tail -5 $O/${TAG}_bench_generator.err
python - $O/${TAG}_bench_mtx.json $O/${TAG}_bench_generator.json <<'PY'
import json, sys
for f in sys.argv[1:]:
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        r = d["roofline"]
        print(f'{d["data"][:60]:60s} {d["value"]:8.1f} GFLOP/s {d["ms_per_step"] * 1e3:7.2f} us  frac {r["frac"]} ({r["bytes_basis"][:40]})  parity {d["parity"]["rows_over_1e-12"]}')
    except Exception as e:
        print(f, "unreadable:", e)
PY
rm -rf $W
