#!/usr/bin/env python3
"""bench.py -- fp64 EHYB SpMV throughput on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

A "step" is one SpMV of the whole matrix: y = A x through libehyb.so's HIP kernels (N = 1),
or, for N > 1, one exchange of x entries over RCCL plus each rank's local multiply:
  --scaling weak (default)  the N = 1 matrix once per GPU: N such grids stacked along z, rank r
                            generates, partitions and uploads block r only; per step a halo exchange
                            (all_to_all of the x entries next to the block boundaries) overlapped
                            with the ELL phase, then the residual phase on the received entries;
  --scaling strong          the N = 1 matrix sharded by rows (two-level partition), all-gatherv of x.

Workload at N = 1 (per GPU at N > 1): BASELINE.json configs[1], audikw_1 -- as a
statistically matched synthetic, because no .mtx file exists offline: 943,695 rows, 3 unknowns
per node of a 68x68x69 grid truncated to 314,565 nodes, 27-point node coupling plus hashed
second-shell couplings tuned to audikw_1's 77.65 M entries (82.3 per row), node labels
scrambled so that locality has to come from the partitioner.  `data` says so.

Output: ONE JSON line on rank 0 with the contract's keys plus
  roofline      algorithmic bytes of the dominant kernel / its mean launch time (HIP events)
  cpu_baseline  the CPU oracle (a port of the reference's CPU product) timed on this host.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X data-sheet peak (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 achievable)

WORKLOADS = {
    # name: (generator, args, symmetric, description)
    "audikw_1-like": ("fem3d", (943695, 3, 68, 68, 13500, 1, 1),
                      "synthetic stand-in for audikw_1: 943,695 rows, ~77.7 M entries, 3 dof/node FEM-like, scrambled labels"),
    "banded-4M": ("banded", (1 << 22, 32, 1024), "config 3: block-circulant band, 4,194,304 rows x 32 entries, zero residual"),
    "rmat-24": ("rmat", (24, 1 << 27, 1), "config 5: R-MAT 2^24 rows, 2^27 edge samples"),
    "kkt3d-200": ("kkt3d", (200,), "config 4 stand-in: KKT-like saddle point system on a 200^3 grid"),
    "small": ("fem3d", (120000, 3, 35, 35, 13500, 1, 1), "reduced-size smoke workload (NOT a valid bench result)"),
    "bcsstk17-like": ("fem3d", (10974, 3, 62, 59, 250000, 1, 17),
                      "config 1's size (bcsstk17: 10,974 rows, ~430 k entries): the reference's small-matrix branch"),
    "rmat-22": ("rmat", (22, 1 << 25, 1), "R-MAT 2^22 rows, 2^25 edge samples (scaled config 5)"),
    "kkt3d-110": ("kkt3d", (110,), "KKT-like saddle point system on a 110^3 grid (scaled config 4)"),
}


SYMMETRIC_GENERATORS = ("fem3d", "kkt3d", "stencil2d")  # A == A^T by construction (include/ehyb.h)
SYM_MIN_ROWS = 32768  # EHYB_SYM_MIN_ROWS (include/ehyb.h): below it plain storage is faster


def owned_cpus():
    """CPUs this job may really use: the affinity mask, capped by the cgroup CPU quota (v2, then v1)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        try:
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0 and period > 0:
                n = min(n, max(1, -(-quota // period)))
        except (OSError, ValueError):
            pass
    return max(1, n)


def symmetric_storage_pays(gen, gargs):
    """What `--sym-pairs auto` means: a symmetric generator and at least EHYB_SYM_MIN_ROWS rows."""
    rows = {"kkt3d": lambda a: 2 * a[0] ** 3, "stencil2d": lambda a: a[0] * a[1]}.get(gen, lambda a: a[0])(gargs)
    return gen in SYMMETRIC_GENERATORS and rows >= SYM_MIN_ROWS


def timed_steps(step, args, torch, dist, world, dev):
    """W untimed steps, then exactly K steps between barrier + synchronize; max over ranks."""
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
        torch.cuda.synchronize()
    t_start = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
        torch.cuda.synchronize()
    elapsed = time.perf_counter() - t_start
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed


def run_weak(args, E, torch, dist, rank, world, dev, cfg, log, stage_on_cpu):
    """N > 1, weak scaling: the N = 1 matrix once per GPU -- N audikw_1-like grids stacked along z,
    rank r owning (and generating) block r only -- and a halo exchange of the x entries of the two
    grid layers next to each block boundary.  Per-GPU work is the N = 1 workload plus ~1 % coupling."""
    import numpy as np

    from ehyb_spmv_gpu_amd import dist as D
    from oracle import oracle as O

    gen, gargs, desc = WORKLOADS[args.workload]
    n_loc = gargs[0]
    t0 = time.time()
    mr = E.Matrix.generate("fem3d_block", *gargs, rank, world, cfg=cfg)
    I, J, V = mr.I.copy(), mr.J.copy(), mr.V.copy()
    mr.free()
    n_glob = n_loc * world
    cuts = [n_loc * r for r in range(world + 1)]
    r0, r1 = cuts[rank], cuts[rank + 1]
    log(f"[bench] rank 0 generated its block: {n_loc} rows, {len(V)} entries in {time.time() - t0:.1f}s")
    # checker (not timed): the oracle on this rank's rows; x is a function of the global index
    x = E.x_glibc(n_glob)
    y_cpu = O.spmv_coo(n_glob, I, J, V, x)[r0:r1]
    scale = O.abs_rowsum(n_glob, I, J, V, x)[r0:r1]
    t0 = time.time()
    L = D.RankLocalMatrix(I, J, V, cuts, rank, cfg, symmetric=True)  # partition + permute the diagonal block, ghost slots
    del I, J
    sh = D.HaloSpmv(L, dev, overlap=not args.no_overlap, stage_on_cpu=stage_on_cpu)
    sh.set_x_local(x[r0:r1])
    st = sh.plan.stats
    log(f"[bench] rank 0: reorder + plan in {time.time() - t0:.1f}s: ell {st['nnz_ell']} residual {st['nnz_er']} "
        f"ghost slots {L.n_ghost} (receives from {int((L.recv_counts > 0).sum())} ranks)")
    elapsed = timed_steps(sh.step, args, torch, dist, world, dev)
    # parity of what was just timed, every rank on its own rows
    bad, worst = O.check_tolerance(sh.y_local(), y_cpu, scale)
    tot = torch.tensor([float(bad), float(len(V)), float(L.n_ghost)], dtype=torch.float64, device=dev)
    mx = torch.tensor([float(worst), float(L.n_ghost)], dtype=torch.float64, device=dev)
    dist.all_reduce(tot)
    dist.all_reduce(mx, op=dist.ReduceOp.MAX)
    bad, nnz, worst = int(tot[0].item()), int(tot[1].item()), float(mx[0].item())
    log(f"[bench] parity vs CPU oracle (all ranks, own rows): {bad} rows over 1e-12, worst {worst:.3e}")
    if bad:
        raise SystemExit("bench.py: GPU result differs from the CPU oracle; refusing to report a number")
    if rank == 0:
        ms = elapsed / args.steps * 1e3
        out = {
            "metric": "fp64 SpMV GFLOP/s (2*nnz/t_iter), EHYB on MI355X",
            "value": round(2.0 * nnz * args.steps / elapsed / 1e9, 2), "unit": "GFLOP/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 5), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": f"synthetic: {world} x ({desc}), stacked along z, one block per GPU",
            "config": {"workload": args.workload, "rows": n_glob, "nnz": nnz, "rows_per_gpu": n_loc,
                       "lds_doubles": int(cfg.lds_doubles), "part_rows": int(cfg.part_rows), "threads": int(cfg.threads),
                       "window_mode": "halo" if cfg.window_mode != 1 else "reference",
                       "sym_pairs_rank0": st["sym_pairs"], "stored_values_rank0": st["size_block_ell"],
                       "ghost_slots_per_gpu_max": int(mx[1].item()), "ghost_slots_total": int(tot[2].item()),
                       "exchange": "halo: gather of the requested x entries + RCCL all_to_all_single into the ghost slots, "
                                   "overlapped with the ELL phase" if not stage_on_cpu else "halo via gloo point-to-point (functional mode)"},
            "alg_GBps": round((12 * nnz + 4 * (n_glob + 1) + 16 * n_glob) / (elapsed / args.steps) / 1e9, 1),
            "roofline": None, "cpu_baseline": None, "parity": {"rows_over_1e-12": bad, "worst_rel": worst},
        }
        print(json.dumps(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="audikw_1-like", choices=sorted(WORKLOADS))
    ap.add_argument("--lds-doubles", type=int, default=0)
    ap.add_argument("--part-rows", type=int, default=0)
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--items-per-cu", type=int, default=0)
    ap.add_argument("--window-mode", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-overlap", action="store_true", help="N>1: exchange first, then multiply (no stream overlap)")
    ap.add_argument("--no-plain-arm", action="store_true",
                    help="N=1 with symmetric pair storage: skip the extra plain-storage measurement of the same matrix")
    ap.add_argument("--sym-pairs", default="auto", choices=["auto", "on", "off"],
                    help="symmetric pair storage (cfg.sym_pairs): auto = on for the symmetric workloads")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="N>1: weak = the N=1 matrix once per GPU, rank-local build, halo exchange (fem3d workloads); "
                         "strong = the N=1 matrix sharded by rows, all-gatherv of x")
    ap.add_argument("--verbose", action="store_true")
    args = ap.parse_args()

    import numpy as np
    import torch

    import ehyb_spmv_gpu_amd as E

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the EHYB multiply has no CPU fallback")
    # EHYB_BENCH_ONE_DEVICE=1 + EHYB_BENCH_BACKEND=gloo: all ranks on cuda:0 over gloo -- a functional
    # test of the N > 1 path on a one-GPU box (not a measurement; RCCL refuses shared devices).
    one_device = os.environ.get("EHYB_BENCH_ONE_DEVICE") == "1"
    backend = os.environ.get("EHYB_BENCH_BACKEND", "nccl")
    if one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    def log(*a):
        if rank == 0:
            print(*a, file=sys.stderr, flush=True)

    kw = {}
    for k, v in (("lds_doubles", args.lds_doubles), ("part_rows", args.part_rows), ("threads", args.threads),
                 ("items_per_cu", args.items_per_cu), ("window_mode", args.window_mode)):
        if v:
            kw[k] = v
    if world > 1 and os.environ.get("OMP_NUM_THREADS") == "1":
        # torch.distributed.run pins every rank to one OpenMP thread; the host pre-step (partitioner,
        # layout builder) of each rank gets its share of the CPUs the job owns instead (affinity mask and
        # cgroup quota -- a container may see many more hardware threads than it may use)
        kw["host_threads"] = max(1, owned_cpus() // world)
    gen, gargs, desc = WORKLOADS[args.workload]
    weak = world > 1 and args.scaling == "weak" and gen == "fem3d"
    # Symmetric pair storage for matrices that are symmetric (the reference reads such files with
    # matrixRead_sym, solver_test.c:127-265, and knows it too): an in-partition pair is stored once.
    sym = args.sym_pairs == "on" or (args.sym_pairs == "auto" and symmetric_storage_pays(gen, gargs))
    if sym:
        kw["sym_pairs"] = 1
    cfg = E.make_config(n_top=1 if weak else world, verbose=1 if (args.verbose and rank == 0) else 0, **kw)

    if weak:  # every rank partitions its own block
        run_weak(args, E, torch, dist, rank, world, torch.device("cuda", local_rank), cfg, log, stage_on_cpu=backend != "nccl")
        dist.destroy_process_group()
        return
    t0 = time.time()
    m = E.Matrix.generate(gen, *gargs, cfg=cfg)
    n, nnz = m.n, m.nnz
    log(f"[bench] generated {args.workload}: n={n} nnz={nnz} in {time.time() - t0:.1f}s")

    # ---- CPU baseline on the un-permuted matrix (rank 0, N = 1 only): the oracle, timed
    cpu_baseline = None
    x = E.x_glibc(n)
    y_cpu = scale = None
    if rank == 0 and world > 1:
        from oracle import oracle as O

        y_cpu = O.spmv_coo(n, m.I, m.J, m.V, x)  # checker for the sharded result (not timed)
        scale = O.abs_rowsum(n, m.I, m.J, m.V, x)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle as O

        rowptr = m.row_idx.astype(np.int64)
        O.set_threads(E.host_threads())  # the CPUs this process owns (cgroup quota), not the ones it sees
        t_coo, y_cpu = O.time_spmv(0, rowptr, m.I, m.J, m.V, x, reps=3)
        t_csr1, _ = O.time_spmv(1, rowptr, m.I, m.J, m.V, x, reps=3)
        t_omp, _ = O.time_spmv(2, rowptr, m.I, m.J, m.V, x, reps=5)
        cores = O.max_threads()
        cpu_baseline = {
            "value": round(2.0 * nnz / t_omp / 1e9, 3), "unit": "GFLOP/s", "cores": cores, "kind": "port",
            "sample": (f"whole {args.workload} matrix, best of 5 CSR fp64 SpMVs with OpenMP on {cores} threads; "
                       f"1-thread CSR {2.0 * nnz / t_csr1 / 1e9:.3f} GFLOP/s; literal reference order "
                       f"(solver_test.c:102, 1 thread) {2.0 * nnz / t_coo / 1e9:.3f} GFLOP/s"),
        }
        scale = O.abs_rowsum(n, m.I, m.J, m.V, x)
        log(f"[bench] cpu baseline: {cpu_baseline['value']} GFLOP/s on {cores} threads")

    # ---- host pre-step: partition + permute (matrixReorder), then this rank's plan
    t0 = time.time()
    m.reorder(cfg)
    log(f"[bench] reorder (partition into {m.c.nParts} parts) {time.time() - t0:.1f}s")
    perm = m.reorder_list.copy()
    xp = E.vector_reorder(x, perm)
    from ehyb_spmv_gpu_amd import dist as D

    dev = torch.device("cuda", local_rank)
    t0 = time.time()
    sh = D.ShardedSpmv(m, cfg, rank, world, dev, overlap=not args.no_overlap)   # plan for this rank's row block
    sh.set_x(xp)
    plan, r0, r1, row_cuts = sh.plan, sh.r0, sh.r1, sh.cuts
    x_d, y_d = sh.x, sh.y
    st = plan.stats
    log(f"[bench] rank {rank}: rows [{r0},{r1}) plan built+uploaded in {time.time() - t0:.1f}s: "
        f"ell {st['nnz_ell']} er {st['nnz_er']} pad {st['ell_padding']} items {st['n_items']} lds {st['lds_bytes']}B")
    stream = torch.cuda.current_stream().cuda_stream
    step = sh.step  # N = 1: one SpMV; N > 1: all-gatherv of the x segments over xGMI + local multiply

    elapsed = timed_steps(step, args, torch, dist, world, dev)

    # ---- parity of what was just timed (rank-local rows) against the CPU oracle
    parity = None
    if world > 1:
        # every rank holds its own y rows: collect all segments (same exchange as for x)
        D.exchange_segments(y_d, row_cuts, rank)
        torch.cuda.synchronize()
    if y_cpu is not None:
        from oracle import oracle as O

        y = E.vector_recover(y_d.cpu().numpy(), perm)
        bad, worst = O.check_tolerance(y, y_cpu, scale)
        parity = {"rows_over_1e-12": bad, "worst_rel": worst}
        log(f"[bench] parity vs CPU oracle: {bad} rows over 1e-12, worst {worst:.3e}")
        if bad:
            raise SystemExit("bench.py: GPU result differs from the CPU oracle; refusing to report a number")

    # ---- per-kernel timing with HIP events on the launch stream (N = 1)
    roofline = None
    if world == 1:
        r = plan.bench(x_d.data_ptr(), y_d.data_ptr(), stream, warmup=5, iters=min(args.steps, 200))
        ell_ms, er_ms = r["ms_ell_avg"], r["ms_er_avg"]
        inline = st["er_inline"] > 0  # the ELL launch also multiplies the (tiny) residual: one launch per SpMV
        empty = st["nnz_er"] == 0
        if inline or empty:
            er_ms = 0.0  # no residual launch: the interval between the two events is event overhead
        bytes_ell = 12 * (st["nnz_ell"] + (st["nnz_er"] if inline else 0)) + 4 * (st["n_rows"] + 1) + 8 * st["n_cols"] + 8 * st["n_rows"]
        achieved = bytes_ell / (ell_ms * 1e-3) / 1e9
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_latest.json")
        if os.path.exists(pmc):
            try:
                traffic = json.load(open(pmc)).get("ehyb_ell_kernel_hbm_bytes_per_launch")
            except Exception:
                traffic = None
        roofline = {"bound": "hbm", "kernel": "ehyb_ell_kernel", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS,
                    "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic,
                    "alg_bytes_per_launch": bytes_ell, "avg_launch_ms": round(ell_ms, 5),
                    "er_kernel_avg_launch_ms": None if (inline or empty) else round(er_ms, 5),
                    "residual": "empty" if empty else ("inline in the ELL launch" if inline else "own launch"),
                    "format_bytes_per_spmv": st["bytes_format"],
                    "whole_spmv_alg_GBps": round(st["bytes_alg"] / ((ell_ms + er_ms) * 1e-3) / 1e9, 1)}
        if traffic:
            roofline["hbm_GBps_from_traffic"] = round(traffic / (ell_ms * 1e-3) / 1e9, 1)
        if st["sym_pairs"] > 0:
            roofline["note"] = ("symmetric pair storage: %d of the %d entries are in-partition pairs a_ij == a_ji stored once "
                                "(one value read, two FMAs, the mirror product added in LDS), so the algorithmic rate "
                                "(12 B per entry, SURVEY 8d) can exceed the HBM peak; traffic and hbm_GBps_from_traffic "
                                "are the bytes really moved" % (2 * st["sym_pairs"], st["nnz"]))

    # ---- the same matrix with plain storage (every entry stored, as the reference does), for the record
    plain = None
    if world == 1 and st["sym_pairs"] > 0 and not args.no_plain_arm:
        t0 = time.time()
        cfg_p = E.make_config(verbose=0, **{k: v for k, v in kw.items() if k != "sym_pairs"})
        mp = E.Matrix.generate(gen, *gargs, cfg=cfg_p)
        mp.reorder(cfg_p)
        plan_p = E.Plan(mp, cfg_p)
        xp_d = E.DeviceBuffer(n).upload(E.vector_reorder(x, mp.reorder_list))
        yp_d = E.DeviceBuffer(n)
        rp_ = plan_p.bench(xp_d.ptr, yp_d.ptr, warmup=args.warmup, iters=args.steps)
        ms_p = rp_["ms_total"] / args.steps
        stp = plan_p.stats
        bytes_p = 12 * (stp["nnz_ell"] + (stp["nnz_er"] if stp["er_inline"] else 0)) + 4 * (n + 1) + 16 * n
        plain = {"value": round(2.0 * nnz / ms_p / 1e6, 2), "unit": "GFLOP/s", "ms_per_step": round(ms_p, 5),
                 "ell_kernel_avg_launch_ms": round(rp_["ms_ell_avg"], 5),
                 "roofline_frac": round(bytes_p / (rp_["ms_ell_avg"] * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                 "stored_values": stp["size_block_ell"], "format_bytes_per_spmv": stp["bytes_format"],
                 "note": "every entry stored (bench.py --sym-pairs off); same run, same GPU"}
        if y_cpu is not None:
            from oracle import oracle as O

            badp, worstp = O.check_tolerance(E.vector_recover(yp_d.download(), mp.reorder_list), y_cpu, scale)
            plain["parity"] = {"rows_over_1e-12": badp, "worst_rel": worstp}
            if badp:
                raise SystemExit("bench.py: plain-storage result differs from the CPU oracle")
        log(f"[bench] plain-storage arm: {plain['value']} GFLOP/s ({time.time() - t0:.1f}s incl. its own pre-step)")
        plan_p.destroy()

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = 2.0 * nnz * args.steps / elapsed / 1e9
        out = {
            "metric": "fp64 SpMV GFLOP/s (2*nnz/t_iter), EHYB on MI355X",
            "value": round(value, 2), "unit": "GFLOP/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 5), "higher_is_better": True,
            "scaling": "strong" if world > 1 else args.scaling, "vs_baseline": None,
            "dtype": "f64", "data": "synthetic: " + desc,
            "config": {"workload": args.workload, "rows": n, "nnz": nnz, "parts": int(m.c.nParts),
                       "lds_doubles": int(cfg.lds_doubles), "part_rows": int(cfg.part_rows), "threads": int(cfg.threads),
                       "window_mode": "halo" if cfg.window_mode != 1 else "reference",
                       "nnz_ell": st["nnz_ell"], "nnz_er": st["nnz_er"], "ell_padding": st["ell_padding"],
                       "sym_pairs": st["sym_pairs"], "stored_values": st["size_block_ell"],
                       "alg_bytes_per_spmv": st["bytes_alg"] if world == 1 else None,
                       "exchange": "none" if world == 1 else "RCCL all-gatherv of x segments"},
            "alg_GBps": round((12 * nnz + 4 * (n + 1) + 16 * n) / (elapsed / args.steps) / 1e9, 1),
            "roofline": roofline, "cpu_baseline": cpu_baseline, "parity": parity,
        }
        if plain:
            out["plain_storage"] = plain
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
