set -x
cd $GRAFT_REPO_ROOT
python bench.py --steps 200 --warmup 20 > gpurun_out/f_bench.json 2> gpurun_out/f_bench.err; tail -1 gpurun_out/f_bench.json | cut -c1-400
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/f_prof -- python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/f_bench_under_rocprof.json 2> gpurun_out/f_prof.err
find gpurun_out/f_prof -name "*kernel_stats.csv" | head -3
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/f_pmc_fetch -- python3 tools/pmc_run.py > gpurun_out/f_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/f_pmc_write -- python3 tools/pmc_run.py > gpurun_out/f_pmc_write.log 2>&1
python tools/pmc_parse.py gpurun_out/f_pmc_fetch gpurun_out/f_pmc_write gpurun_out/f_pmc.json > /dev/null; grep -n "ehyb_ell_kernel_hbm\|factor" gpurun_out/f_pmc.json
EHYB_BENCH_ONE_DEVICE=1 EHYB_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 20 --warmup 3 2>gpurun_out/f_weak2.err | tail -1 | cut -c1-900
