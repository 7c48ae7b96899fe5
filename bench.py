#!/usr/bin/env python3
"""bench.py -- fp64 EHYB SpMV throughput on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

A "step" is one SpMV of the whole matrix, y = A x, through libehyb.so's HIP kernels.

N = 1  workload = BASELINE.json configs[1], audikw_1 -- as a statistically matched synthetic, because
       no .mtx file exists offline: 943,695 rows, 3 unknowns per node of a 68x68x69 grid truncated to
       314,565 nodes, 27-point node coupling plus hashed second-shell couplings tuned to audikw_1's
       77.65 M entries (82.3 per row), node labels scrambled so that locality has to come from the
       partitioner.  `data` says so.  The same line also carries
         plain_storage   the same matrix with every entry stored (no symmetric pair storage),
         dropin_path     the same matrix through the reference-named calls with the reference driver's
                         own sizing (matrixReorder -> vectorReorder -> spmvGPuEHYB -> vectorRecover),
         scaling_anchor  the N > 1 default workload (R-MAT 2^24, config 5) on this one GPU, i.e. the
                         N = 1 point of the strong-scaling curve.
N > 1  one process per GPU (the driver launches them with torch.distributed.run; invoked plainly,
       `python bench.py --gpus N` starts them itself as a child process before anything touches a GPU).
         --scaling strong (default)  ONE matrix (default: config 5, R-MAT 2^24 rows / 2^27 samples)
               sharded by rows over the GPUs: two-level partition (nnz-balanced row blocks, then
               window-sized partitions inside each), per-GPU y segments, and every step an exchange of x
               over RCCL/xGMI: --exchange halo (default) = only the entries a GPU's rows reference, one
               all_to_all_single; --exchange allgather = every segment to everyone, padded to equal
               length, one all_gather_into_tensor.  Phase 1 (LDS-fed ELL part, local columns only)
               overlaps the exchange.
         --scaling weak             the N = 1 matrix once per GPU (fem3d workloads), rank-local build,
               halo exchange.

Output: ONE JSON line on rank 0 with the contract's keys plus
  roofline      the dominant kernel (roofline_block): frac = the counters' bytes per launch (rocprofv3 --pmc FETCH_SIZE /
                WRITE_SIZE taken by this run, else the table entry of exactly this layout, else the format's bytes)
                / the launch's share of one multiply in the graph-replayed loop (HIP events on the launch stream)
                / 8 TB/s.  The counters see what the L2s request from the fabric -- HBM and Infinity-Cache hits alike --
                so the rate is reported as fabric_GBps; frac_first_to_last is the same bytes on the cold-cache clock
                (cfg.ell_alternate = 2), alg_frac SURVEY 8d's 12 nnz + 4 (rows + 1) + 8 cols + 8 rows on the same clock
  cpu_baseline  the CPU oracle (a port of the reference's CPU product) timed on this host.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X data-sheet peak (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 achievable)

WORKLOADS = {
    # name: (generator, args, description)
    "audikw_1-like": ("fem3d", (943695, 3, 68, 68, 13500, 1, 1),
                      "synthetic stand-in for audikw_1: 943,695 rows, ~77.7 M entries, 3 dof/node FEM-like, scrambled labels"),
    "audikw_1-graded": ("fem3d_graded", (943695, 3, 68, 68, 100000, 705000, 1, 1),
                        "harder stand-in for audikw_1: same size (943,695 rows, 77.65 M entries) on a GRADED mesh -- rows of 15 to 303 entries "
                        "(mean 82.3, sigma 41; audikw_1: 21 to 345, mean 82.3), scrambled labels"),
    "audikw_1-mesh": ("mesh3d", (943695, 3, 24, 1500, 1),
                      "unstructured stand-in for audikw_1: 314,565 random points (denser towards one corner), 24 nearest neighbours each, "
                      "made symmetric, 3 unknowns per node -- 943,695 rows, ~78 M entries, rows of 75 to ~150 entries, no lattice anywhere"),
    "banded-4M": ("banded", (1 << 22, 32, 1024), "config 3: block-circulant band, 4,194,304 rows x 32 entries, zero residual"),
    "rmat-24": ("rmat", (24, 1 << 27, 1), "config 5: R-MAT 2^24 rows, 2^27 edge samples"),
    "kkt3d-200": ("kkt3d", (200,), "config 4 stand-in: KKT-like saddle point system on a 200^3 grid"),
    "small": ("fem3d", (120000, 3, 35, 35, 13500, 1, 1), "reduced-size smoke workload (NOT a valid bench result)"),
    "bcsstk17-like": ("fem3d", (10974, 3, 62, 59, 250000, 1, 17),
                      "config 1's size (bcsstk17: 10,974 rows, ~430 k entries): the reference's small-matrix branch"),
    "rmat-22": ("rmat", (22, 1 << 25, 1), "R-MAT 2^22 rows, 2^25 edge samples (scaled config 5)"),
    "rmat-18": ("rmat", (18, 1 << 21, 1), "R-MAT 2^18 rows, 2^21 edge samples (functional tests of the N > 1 path)"),
    "kkt3d-110": ("kkt3d", (110,), "KKT-like saddle point system on a 110^3 grid (scaled config 4)"),
}
# the reference driver's own sizing for audikw_1 (solver_test.c:158-182 with n = 943,695): what a
# drop-in caller hands to matrixReorder / spmvGPuEHYB
REFERENCE_SIZING = {"audikw_1-like": (164, 6144, 0)}

SYMMETRIC_GENERATORS = ("fem3d", "fem3d_graded", "kkt3d", "stencil2d", "mesh3d")  # A == A^T by construction (include/ehyb.h)
SYM_MIN_ROWS = 45056  # EHYB_SYM_MIN_ROWS (include/ehyb.h): below it the direct shape with every entry stored is faster
PMC_FILE = os.path.join(ROOT, "profiles", "pmc_traffic.json")


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


LAYOUT_KEYS = ("nnz", "size_block_ell", "nnz_er", "er_partials", "n_items", "bytes_format")


def layout_fingerprint(st):
    """What ties a PMC measurement to the layout it was taken on: the stored values, the residual's size and form, the
    work items and the format's byte count of a plan (ehyb_plan_stats)."""
    return {k: int(st[k]) for k in LAYOUT_KEYS}


def pmc_traffic(workload, sym, kernel, st=None):
    """HBM bytes per launch measured with rocprofv3 --pmc for exactly this workload, storage and kernel
    (tools/pmc_parse.py writes the table) AND, when the plan's statistics are given, for exactly this layout: an entry
    taken on other partitions / another residual form / other work items is stale and is not quoted.  -> bytes or None."""
    try:
        tab = json.load(open(PMC_FILE))
    except (OSError, ValueError):
        return None
    e = tab.get("entries", {}).get(f"{workload}|{'sym' if sym else 'plain'}|{kernel}")
    if not e:
        return None
    if st is not None and e.get("layout") != layout_fingerprint(st):
        return None
    return e.get("hbm_bytes_per_launch")


def live_pmc_traffic(workload, sym, kname, st, log, timeout_s=300, mtx=""):
    """HBM bytes per launch of the dominant kernel MEASURED BY THIS RUN, not looked up: two child processes -- tools/pmc_run.py
    (the same generator, reorder and plan, a 1 GiB streaming read for the calibration of FETCH_SIZE, ten multiplies) under
    `rocprofv3 --pmc FETCH_SIZE --kernel-trace` and `--pmc WRITE_SIZE --kernel-trace`, separate passes as MI355X_MICROARCH.md
    prescribes -- parsed by tools/pmc_parse.py.  Quoted only if the child's plan has this plan's layout fingerprint.
    -> (bytes or None, what happened)"""
    import shutil
    import tempfile

    if any("rocprof" in os.environ.get(k, "").lower() for k in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "HSA_TOOLS_LIB")):
        return None, "this process runs under a profiler itself"
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        return None, "rocprofv3 not found"
    if workload not in WORKLOADS and not mtx:
        return None, "not a generator workload"
    tools = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools")
    tmp = tempfile.mkdtemp(prefix="ehyb_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    t0 = time.time()
    try:
        for counter, sub in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
            cmd = [rocprof, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", os.path.join(tmp, sub), "--",
                   sys.executable, os.path.join(tools, "pmc_run.py")] + (["--mtx", os.path.abspath(mtx)] if mtx else ["--workload", workload]) + ([] if sym else ["--plain"])
            if sub == "fetch":
                cmd += ["--layout-out", os.path.join(tmp, "layout.json")]
            rc, err = run_in_own_group(cmd, env, "/tmp", timeout_s)
            if rc != 0:
                return None, f"rocprofv3 --pmc {counter} failed ({rc}): {err[-200:]}"
        out = os.path.join(tmp, "traffic.json")
        p = subprocess.run([sys.executable, os.path.join(tools, "pmc_parse.py"), os.path.join(tmp, "fetch"), os.path.join(tmp, "write"), out,
                            "--workload", workload, "--storage", "sym" if sym else "plain", "--layout", os.path.join(tmp, "layout.json")],
                           capture_output=True, text=True, timeout=120)
        if p.returncode != 0:
            return None, "pmc_parse.py failed: " + p.stderr[-200:]
        res = json.load(open(out))
        if res.get("layout") != layout_fingerprint(st):
            return None, "the profiled child built another layout (bench options the child does not take)"
        parts = [k for name, k in res["kernels"].items() if ("ehyb_pb_" in name if kname.startswith("ehyb_pb_") else kname in name) and k.get("hbm_bytes_per_launch")]
        want = 2 if kname.startswith("ehyb_pb_") else 1
        if len(parts) != want:
            return None, f"{len(parts)} kernels named {kname} in the counter files"
        total = float(sum(k["hbm_bytes_per_launch"] for k in parts))
        factor = (res.get("fetch_calibration") or {}).get("factor")
        if not factor:
            return None, "no calibration kernel in the counter files (FETCH_SIZE uncalibrated: not quoted)"
        log(f"[bench] PMC traffic of {kname} measured by this run: {total / 1e6:.2f} MB per launch (FETCH_SIZE x {factor:.5f} + WRITE_SIZE; {time.time() - t0:.1f}s)")
        return total, {"fetch_factor": factor, "launches": min(k["launches"] for k in parts), "seconds": round(time.time() - t0, 1)}
    except Exception as e:  # noqa: BLE001  (a side diagnostic must never cost the headline its line)
        return None, f"{type(e).__name__}: {e}"
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def run_in_own_group(cmd, env, cwd, timeout_s):
    """A child in its own process group, so that a timeout ends the profiler AND the program it started (the group this
    call created, nothing else).  -> (return code, tail of stderr)"""
    import signal

    p = subprocess.Popen(cmd, env=env, cwd=cwd, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True, start_new_session=True)
    try:
        _, err = p.communicate(timeout=timeout_s)
        return p.returncode, err or ""
    except subprocess.TimeoutExpired:
        try:
            os.killpg(p.pid, signal.SIGKILL)
        except OSError:
            pass
        p.wait()
        return -9, f"timed out after {timeout_s}s"


def self_launch(args):
    """`python bench.py --gpus N` invoked plainly: start the N ranks as a CHILD process
    (torch.distributed.run) before this process has touched a GPU, relay rank 0's JSON line."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p = subprocess.run(cmd, env=env)
    raise SystemExit(p.returncode)


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
    t_issued = time.perf_counter()   # the host has ENQUEUED every step (it runs ahead of the device unless it is the slower one)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
        torch.cuda.synchronize()
    elapsed = time.perf_counter() - t_start
    args.host_us_per_step = (t_issued - t_start) / args.steps * 1e6
    if world > 1:
        t = torch.tensor([elapsed, args.host_us_per_step], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, args.host_us_per_step = float(t[0].item()), float(t[1].item())
    return elapsed


def all_ranks_agree_or_exit(bad, worst, torch, dist, world, dev, what):
    """Every rank learns the parity verdict of every other and all leave together."""
    if world > 1:
        t = torch.tensor([float(bad)], dtype=torch.float64, device=dev)
        w = torch.tensor([float(worst)], dtype=torch.float64, device=dev)
        dist.all_reduce(t)
        dist.all_reduce(w, op=dist.ReduceOp.MAX)
        bad, worst = int(t.item()), float(w.item())
    if bad:
        if world > 1:
            dist.destroy_process_group()
        raise SystemExit(f"bench.py: {what}: GPU result differs from the CPU oracle in {bad} rows (worst {worst:.3e}); refusing to report a number")
    return bad, worst


def partitioner_for(E, gen):  # ("file": a real matrix -- EHYB_PART_AUTO finds out by itself)
    """R-MAT has no locality for a graph partitioner to find (2^24 rows: 124 M of 133 M edges cut after
    110 s of multilevel k-way); EHYB_PART_AUTO notices that itself after one coarsening attempt and falls back to
    blocks of the degree order -- naming that order up front saves even the attempt.  (Round 2 cut contiguous blocks
    of the generator's own numbering: the degree order leaves the panel-form residual 35-45 % fewer partial sums.)"""
    return E.EHYB_PART_DEGREE if gen == "rmat" else E.EHYB_PART_AUTO


def step_comm(args, D, dist, stage_on_cpu, log):
    """The communicator of the C-side step (--step): libehyb.so's own RCCL communicator, made from a ncclUniqueId that
    torch.distributed broadcasts.  -> (Comm or None, what the JSON line says).  `--step auto` falls back to the Python-issued
    collectives when the communicator cannot be made on EVERY rank (an all-reduced verdict: no rank goes on alone), and says so;
    `--step c` fails instead."""
    import torch

    if args.step == "python" or (args.step == "auto" and stage_on_cpu):
        return None, "python: torch.distributed collectives, one C call per part" + (" (gloo functional mode)" if stage_on_cpu else "")
    if stage_on_cpu:
        raise SystemExit("bench.py: --step c needs the nccl backend (RCCL refuses ranks that share a device)")
    comm, err = None, ""
    try:
        comm = D.make_comm()
    except Exception as e:  # noqa: BLE001
        err = f"{type(e).__name__}: {e}"
    ok = torch.tensor([1.0 if comm is not None else 0.0], dtype=torch.float64, device=torch.device("cuda", torch.cuda.current_device()))
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if ok.item() < 1.0:
        if comm is not None:
            comm.destroy()
        if args.step == "c":
            raise SystemExit(f"bench.py: --step c: the RCCL communicator could not be made on every rank ({err or 'another rank failed'})")
        log(f"[bench] C-side step unavailable ({err or 'another rank failed'}): torch.distributed collectives issued from Python instead")
        return None, f"python (fallback: {err or 'another rank could not join the communicator'})"
    return comm, "c: libehyb.so's RCCL communicator, ONE host call per multiply (ehyb_halo_spmv / ehyb_gather_spmv)"


def run_weak(args, E, torch, dist, rank, world, dev, cfg, log, stage_on_cpu):
    """N > 1, weak scaling: the N = 1 matrix once per GPU -- N audikw_1-like grids stacked along z,
    rank r owning (and generating) block r only -- and a halo exchange of the x entries of the two
    grid layers next to each block boundary.  Per-GPU work is the N = 1 workload plus ~1 % coupling."""
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
    comm, step_impl = step_comm(args, D, dist, stage_on_cpu, log)
    sh = D.HaloSpmv(L, dev, overlap=not args.no_overlap, stage_on_cpu=stage_on_cpu, comm=comm)
    sh.set_x_local(x[r0:r1])
    st = sh.plan.stats
    log(f"[bench] rank 0: reorder + plan in {time.time() - t0:.1f}s: ell {st['nnz_ell']} residual {st['nnz_er']} "
        f"ghost slots {L.n_ghost} (receives from {int((L.recv_counts.sum(axis=0) > 0).sum())} ranks)")
    elapsed = timed_steps(sh.step, args, torch, dist, world, dev)
    # parity of what was just timed, every rank on its own rows
    bad, worst = O.check_tolerance(sh.y_local(), y_cpu, scale)
    bad, worst = all_ranks_agree_or_exit(bad, worst, torch, dist, world, dev, "weak scaling")
    tot = torch.tensor([float(len(V)), float(L.n_ghost)], dtype=torch.float64, device=dev)
    mx = torch.tensor([float(L.n_ghost)], dtype=torch.float64, device=dev)
    dist.all_reduce(tot)
    dist.all_reduce(mx, op=dist.ReduceOp.MAX)
    nnz = int(tot[0].item())
    log(f"[bench] parity vs CPU oracle (all ranks, own rows): {bad} rows over 1e-12, worst {worst:.3e}")
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
                       "ghost_slots_per_gpu_max": int(mx[0].item()), "ghost_slots_total": int(tot[1].item()),
                       "step_issued_by": step_impl,
                       "exchange": "halo: gather of the requested x entries + RCCL all_to_all_single into the ghost slots, "
                                   "overlapped with the ELL phase" if not stage_on_cpu else "halo via gloo point-to-point (functional mode)"},
            "alg_GBps": round((12 * nnz + 4 * (n_glob + 1) + 16 * n_glob) / (elapsed / args.steps) / 1e9, 1),
            "roofline": None, "cpu_baseline": None, "parity": {"rows_over_1e-12": bad, "worst_rel": worst},
        }
        print(json.dumps(out), flush=True)
    if comm is not None:
        torch.cuda.synchronize()
        comm.destroy()


def run_strong(args, E, torch, dist, rank, world, dev, cfg, log, stage_on_cpu):
    """N > 1, strong scaling (BASELINE config 5): ONE matrix sharded by rows, one block per GPU.
    R-MAT: contiguous blocks with equal numbers of edge samples, every rank generating its own block only
    (ehyb_gen_rmat_block).  Matrices with locality: every rank generates the matrix (deterministic
    generators; a real deployment reads its rows from a file) and takes its block of a k-way graph
    partition balanced on entries.  A rank keeps rows [r0, r1) of its block, builds its
    plan from them alone (window-sized partitions inside: the second level) and exchanges x
    every step: `--exchange allgather` = the x segments, padded to equal length, through one RCCL
    all_gather_into_tensor (what north_star names); `--exchange halo` = only the entries the rank's
    rows reference, through one all_to_all_single.  The ELL phase (local columns) overlaps either."""
    import numpy as np

    from ehyb_spmv_gpu_amd import dist as D
    from oracle import oracle as O

    gen, gargs, desc = WORKLOADS[args.workload]
    if args.exchange is None:
        args.exchange = "cover" if gen == "rmat" else "halo"
    if args.exchange == "cover" and gen != "rmat":
        raise SystemExit("bench.py: --exchange cover builds every rank's plan in panel form: it is for the R-MAT workloads (a mesh's halo is a few per cent of x anyway)")
    t0 = time.time()
    if gen == "rmat":
        # no locality to find: contiguous row blocks with equal numbers of edge samples in the matrix's own
        # numbering; every rank draws all samples twice (histogram, then its own block) and keeps only its rows
        # (the cuts balance what a rank will multiply: under the cover exchange the entries of the hub rows mostly go to the columns' owners)
        m = E.Matrix.generate("rmat_block", *gargs, rank, world, 1 if args.exchange == "cover" else 0, cfg=cfg)
        n, cuts = m.n, m.block_cuts
        t = torch.tensor([float(m.nnz)], dtype=torch.float64, device=dev)
        dist.all_reduce(t)
        nnz = int(t.item())
        log(f"[bench] every rank generated its row block of {args.workload}: n={n} nnz={nnz} (rank 0: {m.nnz}) in {time.time() - t0:.1f}s")
        x = E.x_glibc(n)
    else:
        m = E.Matrix.generate(gen, *gargs, cfg=cfg)
        n, nnz = m.n, m.nnz
        log(f"[bench] every rank generated {args.workload}: n={n} nnz={nnz} in {time.time() - t0:.1f}s")
        x = E.x_glibc(n)
        # top level of the two-level partition: a k-way graph partition into one block of rows
        # per GPU, balanced on entries, so that a GPU's rows reference few columns of the others; the
        # matrix and x are taken into that numbering (every rank computes the same permutation)
        t0 = time.time()
        cfg_top = E.make_config(n_top=world, host_threads=int(cfg.host_threads))
        m.reorder(cfg_top)
        pb = m.part_boundary
        cuts = [int(pb[b]) for b in m.block_first]
        x = E.vector_reorder(x, m.reorder_list)
        log(f"[bench] top-level partition into {world} row blocks in {time.time() - t0:.1f}s")
    rowptr = m.row_idx.astype(np.int64)
    r0, r1 = cuts[rank], cuts[rank + 1]
    k0, k1 = int(rowptr[r0]), int(rowptr[r1])
    I, J, V = m.I[k0:k1].copy(), m.J[k0:k1].copy(), m.V[k0:k1].copy()
    symmetric = gen in SYMMETRIC_GENERATORS
    m.free()
    y_cpu = O.spmv_coo(n, I, J, V, x)[r0:r1]      # checker (not timed): the oracle on this rank's rows
    scale = O.abs_rowsum(n, I, J, V, x)[r0:r1]
    t0 = time.time()
    # default: the hot quarter of every owner's ghost columns first (on R-MAT 2^24 at 8 ranks it carries 73-75 % of a rank's
    # entries: tools/dist_stats.py), the rest while those panels multiply
    shares = [float(v) for v in args.chunk_shares.split(",")] if args.chunk_shares else ([0.25, 0.75] if args.chunks == 2 else None)
    L = D.RankLocalMatrix(I, J, V, cuts, rank, cfg, symmetric=symmetric, exchange=args.exchange, chunks=args.chunks, chunk_shares=shares)
    del I, J
    comm, step_impl = step_comm(args, D, dist, stage_on_cpu, log)
    if args.exchange in ("halo", "cover"):
        sh = D.HaloSpmv(L, dev, overlap=not args.no_overlap, stage_on_cpu=stage_on_cpu, mode=args.exchange_mode, comm=comm)
    else:
        sh = D.GatherSpmv(L, dev, overlap=not args.no_overlap, stage_on_cpu=stage_on_cpu, comm=comm)
    sh.set_x_local(x[r0:r1])
    st = sh.plan.stats
    log(f"[bench] rank 0: rows [{r0},{r1}) reorder + plan in {time.time() - t0:.1f}s: ell {st['nnz_ell']} residual {st['nnz_er']} "
        f"ghost columns {L.n_ghost} of {n - (r1 - r0)} remote")
    elapsed = timed_steps(sh.step, args, torch, dist, world, dev)
    bad, worst = O.check_tolerance(sh.y_local(), y_cpu, scale)
    bad, worst = all_ranks_agree_or_exit(bad, worst, torch, dist, world, dev, "strong scaling")
    # per-rank time of the local multiply alone (no exchange), max over ranks: what is left is the exchange
    if args.exchange in ("halo", "cover"):
        own_ms, sh_local_ms = sh.time_parts(args.steps)   # the part that needs the rank's own columns only, and the whole
    else:
        own_ms, sh_local_ms = 0.0, sh.time_local(args.steps)
    n_partials = int(L.yrecv_counts.sum())       # exchange "cover": partial sums this rank receives per step (0 otherwise)
    stats = torch.tensor([float(L.n_ghost), float(sh_local_ms), float(L.nnz), float(own_ms), float(L.nnz_own_cols), float(n_partials), float(L.nnz_exported)],
                         dtype=torch.float64, device=dev)
    mx = stats.clone()
    dist.all_reduce(stats)
    dist.all_reduce(mx, op=dist.ReduceOp.MAX)
    # who receives how much from whom, per exchange step: rows = receiving rank, chunks x peers doubles each
    vol = torch.zeros(world, L.chunks * world, dtype=torch.float64, device=dev)
    vol[rank] = torch.from_numpy(L.recv_counts.astype(np.float64).reshape(-1)).to(dev)
    dist.all_reduce(vol)
    log(f"[bench] parity vs CPU oracle (all ranks, own rows): {bad} rows over 1e-12, worst {worst:.3e}")
    # ---- the SAME matrix on ONE GPU, in this very run: rank 0 multiplies the whole matrix on its GPU while the others wait,
    # so that the line carries its own N = 1 point (the N = 1 default of bench.py is BASELINE config 2, another matrix)
    single = None
    if not args.no_single_gpu_anchor:
        if rank == 0:
            t0 = time.time()
            kw1 = {"host_threads": owned_cpus()}   # the other ranks wait: every CPU the job owns builds this one plan
            single = side_arm("single-GPU anchor", lambda: one_gpu_case(E, O, np, args.workload, symmetric and symmetric_storage_pays(gen, gargs), kw1,
                                                                        min(args.steps, 50), min(args.warmup, 5), log, want_parity=False), log)
            log(f"[bench] the same matrix on one GPU: {single.get('value')} GFLOP/s, {single.get('ms_per_step')} ms ({time.time() - t0:.1f}s incl. its pre-step)")
        dist.barrier()
    if rank == 0:
        ms = elapsed / args.steps * 1e3
        words = int(stats[0].item() + stats[5].item()) if args.exchange in ("halo", "cover") else sh.seg_len * world * (world - 1)
        out = {
            "metric": "fp64 SpMV GFLOP/s (2*nnz/t_iter), EHYB on MI355X",
            "value": round(2.0 * nnz * args.steps / elapsed / 1e9, 2), "unit": "GFLOP/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 5), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic: " + desc,
            "config": {"workload": args.workload, "rows": n, "nnz": nnz, "rows_per_gpu_max": max(cuts[i + 1] - cuts[i] for i in range(world)),
                       "nnz_per_gpu_max": int(mx[2].item()),
                       "lds_doubles": int(cfg.lds_doubles), "part_rows": int(cfg.part_rows), "threads": int(cfg.threads),
                       "exchange": ("RCCL all_gather_into_tensor of the x segments (padded to %d doubles each), overlapped with the ELL phase" % sh.seg_len)
                       if args.exchange == "allgather" else
                       (args.exchange + ": gather of the requested x entries + one group of ncclSend / ncclRecv pairs per exchange step (all_to_all pattern, issued "
                        "from C on the communicator's own stream) straight into the ghost " if comm is not None else
                        args.exchange + ": gather of the requested x entries + one RCCL all_to_all_single per exchange step straight into the ghost ") +
                       "columns; own-column panels and the panels of the chunks already delivered multiply while the next chunk travels",
                       "exchange_doubles_received_all_gpus": words,
                       "cover": ({"ghost_columns_all_gpus": int(stats[0].item()), "partial_sums_all_gpus": int(stats[5].item()), "entries_handed_to_the_column_owner": int(stats[6].item()),
                                  "what": "per pair of ranks the hub columns of the block travel as x entries, the rest of the block is multiplied by the columns' owner, who ships "
                                          "one partial sum per row (ehyb_halo_set_partials; DESIGN.md 5)"} if args.exchange == "cover" else None),
                       "step_issued_by": step_impl,
                       "exchange_steps": L.chunks, "exchange_mode": ("grouped ncclSend/ncclRecv pairs" if comm is not None else args.exchange_mode) if args.exchange in ("halo", "cover") else ("ncclAllGather" if comm is not None else "all_gather_into_tensor"),
                       "pipelined": bool(args.exchange in ("halo", "cover") and not args.no_overlap),
                       "recv_doubles_by_rank_step_peer": [[[int(v) for v in row.reshape(L.chunks, world)[k].tolist()] for k in range(L.chunks)]
                                                          for row in vol.cpu().numpy()] if world <= 8 else None,
                       "ghost_columns_per_gpu_max": int(mx[0].item()),
                       "functional_mode": "gloo, host-staged" if stage_on_cpu else None},
            "local_multiply_ms_max_over_ranks": round(float(mx[1].item()), 5),
            # what can run while the first exchange step is on the wire: the ELL launch + the panels of the rank's own columns
            "own_columns_part_ms_max_over_ranks": round(float(mx[3].item()), 5),
            "phase1_share_of_local_ms": round(float(mx[3].item()) / max(float(mx[1].item()), 1e-9), 4),
            "own_columns_share_of_entries": round(float(stats[4].item()) / max(float(stats[2].item()), 1.0), 4),
            "host_us_per_step": round(args.host_us_per_step, 1),
            # strong scaling read off ONE run: the whole job against the same matrix on rank 0's GPU alone
            "single_gpu_same_matrix": single,
            "speedup_vs_single_gpu_same_run": (round(2.0 * nnz * args.steps / elapsed / 1e9 / single["value"], 3)
                                               if single and single.get("value") else None),
            "n1_equivalent": ("the N = 1 line's scaling_anchor (same matrix, one GPU)" if args.workload == "rmat-24" else
                              "the N = 1 line's value (same matrix, one GPU)" if args.workload == "audikw_1-like" else
                              f"bench.py --workload {args.workload} (N = 1)"),
            "alg_GBps": round((12 * nnz + 4 * (n + 1) + 16 * n) / (elapsed / args.steps) / 1e9, 1),
            "roofline": None, "cpu_baseline": None, "parity": {"rows_over_1e-12": bad, "worst_rel": worst},
        }
        print(json.dumps(out), flush=True)
    if comm is not None:
        torch.cuda.synchronize()
        comm.destroy()


def one_gpu_case(E, O, np, workload, sym, kw, steps, warmup, log, want_parity=True, y_cpu=None, scale=None, x=None):
    """Generate -> reorder -> plan -> timed loop of `workload` on the current GPU through the plan API.
    -> dict with value, ms_per_step, kernel times, stats, parity."""
    gen, gargs, _ = WORKLOADS[workload]
    kw = dict(kw)
    kw.pop("sym_pairs", None)
    if sym:
        kw["sym_pairs"] = 1
    cfg = E.make_config(partitioner=partitioner_for(E, gen), **kw)
    t0 = time.time()
    m = E.Matrix.generate(gen, *gargs, cfg=cfg)
    n, nnz = m.n, m.nnz
    if x is None:
        x = E.x_glibc(n)
    if want_parity and y_cpu is None:
        y_cpu = O.spmv_coo(n, m.I, m.J, m.V, x)
        scale = O.abs_rowsum(n, m.I, m.J, m.V, x)
    m.reorder(cfg)
    plan = E.Plan(m, cfg)
    t_pre = time.time() - t0
    xd = E.DeviceBuffer(n).upload(E.vector_reorder(x, m.reorder_list))
    yd = E.DeviceBuffer(n)
    plan.tune(xd.ptr, yd.ptr)      # as the headline does (a no-op for plans without a one-round ELL launch)
    r = plan.bench(xd.ptr, yd.ptr, warmup=warmup, iters=steps)
    st = plan.stats
    ms = r["ms_total"] / steps
    out = {"value": round(2.0 * nnz / ms / 1e6, 2), "unit": "GFLOP/s", "ms_per_step": round(ms, 5), "rows": n, "nnz": nnz,
           "ell_kernel_avg_launch_ms": round(r["ms_ell_avg"], 5),
           "er_kernel_avg_launch_ms": round(r["ms_er_avg"], 5) if (st["nnz_er"] and not st["er_inline"]) else None,
           "nnz_ell": st["nnz_ell"], "nnz_er": st["nnz_er"], "stored_values": st["size_block_ell"],
           "format_bytes_per_spmv": st["bytes_format"], "alg_bytes_per_spmv": st["bytes_alg"],
           # what the FORMAT makes the kernels move (an upper bound of the traffic: units of one panel share an XCD's L2) ...
           "format_frac_of_8TBps": round(st["bytes_format"] / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
           # ... SURVEY 8d's algorithmic bytes ...
           "alg_frac_of_8TBps": round(st["bytes_alg"] / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
           "pre_step_s": round(t_pre, 1)}
    # ... and the counters' bytes where profiles/pmc_traffic.json holds a measurement of exactly this layout
    pmc = pmc_traffic(workload, st["sym_pairs"] > 0, dominant_kernel(st, r["ms_ell_avg"], r["ms_er_avg"]), st)
    if pmc and (st["nnz_ell"] == 0 or st["nnz_er"] == 0 or st["er_inline"] > 0):   # one kernel (pair) is the whole multiply
        out["pmc_bytes_per_spmv"] = pmc
        out["pmc_frac_of_8TBps"] = round(pmc / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)
    if want_parity:
        bad, worst = O.check_tolerance(E.vector_recover(yd.download(), m.reorder_list), y_cpu, scale)
        out["parity"] = {"rows_over_1e-12": bad, "worst_rel": worst}
        if bad:
            raise SystemExit(f"bench.py: {workload} ({'symmetric pairs' if sym else 'every entry'}): result differs from the CPU oracle")
    plan.destroy()
    xd.free()
    yd.free()
    m.free()
    return out


def dropin_case(E, O, np, workload, x, y_cpu, scale, steps, log):
    """The same matrix through the reference-named calls, with the reference driver's own sizing:
    matrixReorder(&m) -> vectorReorder -> spmvGPuEHYB(&m, x, y, steps, &it) -> vectorRecover."""
    gen, gargs, _ = WORKLOADS[workload]
    m = E.Matrix.generate(gen, *gargs)
    m.c.nParts, m.c.vectorCacheSize, m.c.kernelPerPart = REFERENCE_SIZING[workload]
    hints = (int(m.c.nParts), int(m.c.vectorCacheSize))
    t0 = time.time()
    m.reorder_dropin()
    t_re = time.time() - t0
    perm = m.reorder_list.copy()
    # spmvGPuEHYB prints its two lines (spmv.cu:82,121) on stdout, which belongs to the JSON line here
    sys.stdout.flush()
    keep = os.dup(1)
    os.dup2(2, 1)
    try:
        y, it, ms = E.spmv_gpu_ehyb(m, E.vector_reorder(x, perm), steps, timing=True)
    finally:
        import ctypes

        ctypes.CDLL(None).fflush(None)  # the C library's own stdout buffer, before fd 1 is put back
        sys.stdout.flush()
        os.dup2(keep, 1)
        os.close(keep)
    bad, worst = O.check_tolerance(E.vector_recover(y, perm), y_cpu, scale)
    if bad:
        raise SystemExit("bench.py: drop-in path: result differs from the CPU oracle")
    out = {"value": round(2.0 * m.nnz * it / (ms * 1e6), 2), "unit": "GFLOP/s", "ms_per_step": round(ms / it, 5),
           "calls": "matrixReorder -> vectorReorder -> spmvGPuEHYB -> vectorRecover (include/reordering.h, include/spmv.h), no configuration",
           "caller_sizing_hints": {"nParts": hints[0], "vectorCacheSize": hints[1], "from": "solver_test.c:158-182 for audikw_1"},
           "nParts_used": int(m.c.nParts), "reorder_s": round(t_re, 1), "parity": {"rows_over_1e-12": bad, "worst_rel": worst}}
    m.free()
    return out


def side_arm(name, fn, log):
    """The extra measurements of the N = 1 line (plain storage, drop-in calls, vendor library, scaling anchor)
    must not cost the headline its line: a failure there -- a parity mismatch included -- is reported under
    the arm's own key."""
    try:
        return fn()
    except (Exception, SystemExit) as err:  # noqa: BLE001
        log(f"[bench] {name} FAILED: {err}")
        return {"error": str(err) or type(err).__name__}


def side_arm_pair(name, fn, log):
    """side_arm for a measurement that returns (value or None, detail): a failure of any kind becomes (None, what happened)."""
    try:
        return fn()
    except (Exception, SystemExit) as err:  # noqa: BLE001
        log(f"[bench] {name} FAILED: {err}")
        return None, f"{type(err).__name__}: {err}"


def alg_bytes_split(st):
    """SURVEY 8d's algorithmic bytes of ONE SpMV, B_alg = 12 nnz + 4 (rows + 1) + 8 cols + 8 rows, split over the launches of
    a multiply so that the shares add up to B_alg: the launch that writes every row of y (the ELL launch -- or the residual
    launch(es) of a plan without one: R-MAT in panel form) carries the row-pointer and vector terms, a residual launch
    BESIDE an ELL launch carries 12 B per entry of its own.  -> (ell share, residual share)"""
    rows, cols = int(st["n_rows"]), int(st["n_cols"])
    vec = 4 * (rows + 1) + 8 * cols + 8 * rows
    inline = st["er_inline"] > 0
    nnz_er = int(st["nnz_er"])
    if inline or nnz_er == 0:
        return 12 * (int(st["nnz_ell"]) + nnz_er) + vec, 0
    if int(st["nnz_ell"]) == 0:
        return 0, 12 * nnz_er + vec          # no entry is multiplied by an ELL launch (every window given up): the residual launches are the whole multiply
    return 12 * int(st["nnz_ell"]) + vec, 12 * nnz_er


def dominant_kernel(st, ell_ms, er_ms):
    """Name of the launch (pair) a multiply spends most of its time in."""
    if st["er_inline"] > 0 or st["nnz_er"] == 0 or er_ms <= ell_ms:
        return "ehyb_ell_kernel"
    return "ehyb_pb_scale_kernel+ehyb_pb_reduce_kernel" if st["er_partials"] > 0 else "ehyb_er_kernel"


def roofline_block(st, ell_ms, er_ms, step_ms, traffic, bytes_basis, first_to_last_step_ms=None):
    """The `roofline` object of the JSON line for the dominant launch of a multiply.
      st        ehyb_plan_stats of the plan
      ell_ms, er_ms   per-launch averages between HIP event pairs (second pass of ehyb_spmv_bench): used for the SHARES only
      step_ms   mean time of one multiply in the graph-replayed loop between two HIP events on the launch stream; the launch
                time the fractions are taken on is its share of THAT, so `frac` follows from the loop, not from a pass that
                pays an event pair per launch
      traffic   bytes per launch from the PMC counters (FETCH_SIZE x calibration + WRITE_SIZE) or None
    frac      = traffic (else format bytes) / launch time / 8 TB/s -- ALWAYS on the counters when they exist.  FETCH_SIZE counts what
                the L2s request from the fabric, whether HBM or the 256 MB Infinity Cache answers (MI355X_MICROARCH.md), so this is a
                FABRIC rate (`fabric_GBps`), an upper bound of the HBM rate; counters below the format's bytes mean L2 hits (`l2_share`).
    alg_frac  = SURVEY 8d's bytes (alg_bytes_split) / launch time / 8 TB/s."""
    inline = st["er_inline"] > 0
    empty = st["nnz_er"] == 0
    if inline or empty:
        er_ms = 0.0   # no residual launch: the interval between the two events is event overhead
    alg_ell, alg_er = alg_bytes_split(st)
    fmt_ell = int(st["bytes_format_ell"])
    fmt_er = int(st["bytes_format"]) - fmt_ell
    kname = dominant_kernel(st, ell_ms, er_ms)
    tot = ell_ms + er_ms
    if kname == "ehyb_ell_kernel":
        share, k_alg, k_fmt, k_event_ms = (ell_ms / tot if tot > 0 else 1.0), alg_ell, fmt_ell, ell_ms
    else:
        share, k_alg, k_fmt, k_event_ms = er_ms / tot, alg_er, fmt_er, er_ms
    k_ms = step_ms * share
    real_bytes = traffic if traffic else k_fmt
    achieved = real_bytes / (k_ms * 1e-3) / 1e9
    out = {"bound": "hbm", "kernel": kname, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
           "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic, "bytes_basis": bytes_basis,
           "fabric_GBps": round(traffic / (k_ms * 1e-3) / 1e9, 1) if traffic else None,
           "fabric_note": "FETCH_SIZE counts L2 misses served by the fabric: HBM and Infinity-Cache hits alike (no counter of this rocprofv3 tells them apart)",
           "l2_share": round(max(0.0, 1.0 - traffic / k_fmt), 4) if (traffic and k_fmt) else None,
           "format_bytes_per_launch": k_fmt, "avg_launch_ms": round(k_ms, 5), "avg_launch_ms_basis": "graph-replayed loop between two HIP events x the launch's share of a multiply",
           "event_bracketed_launch_ms": round(k_event_ms, 5),
           "alg_bytes_per_launch": k_alg, "alg_GBps": round(k_alg / (k_ms * 1e-3) / 1e9, 1),
           "alg_frac": round(k_alg / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
           "ell_kernel_avg_launch_ms": round(step_ms * (ell_ms / tot if tot > 0 else 1.0), 5),
           "er_kernel_avg_launch_ms": None if (inline or empty) else round(step_ms * er_ms / tot, 5),
           "residual": "empty" if empty else ("inline in the ELL launch" if inline else
                                              ("panel form: two launches (x panels, then y blocks in LDS)" if st["er_partials"] > 0 else "own launch (CSR segments)")),
           "format_bytes_per_spmv": int(st["bytes_format"]),
           "whole_spmv_real_frac": round(int(st["bytes_format"]) / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
           "whole_spmv_alg_GBps": round(int(st["bytes_alg"]) / (step_ms * 1e-3) / 1e9, 1)}
    if kname != "ehyb_ell_kernel":
        # what round 3 printed as alg bytes of the residual: 12 B per entry + the 8 B gathered x entry of a CSR walk -- a model of
        # that kernel's traffic, not SURVEY 8d's figure
        out["gather_model_bytes_per_launch"] = 20 * int(st["nnz_er"]) + 16 * int(st["rows_er"])
    if first_to_last_step_ms:
        add_first_to_last(out, first_to_last_step_ms, step_ms)
    return out


def add_first_to_last(roofline, first_to_last_step_ms, step_ms):
    """`frac` is the steady state of a loop of multiplies (cfg.ell_alternate: a launch starts with what the one before it left in the
    Infinity Cache); a single multiply after other work gets the cold-cache time.  Same bytes, the other clock."""
    k_ms = roofline["avg_launch_ms"] * first_to_last_step_ms / step_ms
    b = roofline["traffic"] if roofline["traffic"] else roofline["format_bytes_per_launch"]
    roofline["first_to_last_launch_ms"] = round(k_ms, 5)
    roofline["frac_first_to_last"] = round(b / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)
    roofline["alg_frac_first_to_last"] = round(roofline["alg_bytes_per_launch"] / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--mtx", default="", help="N=1: the headline on a Matrix Market file (e.g. ./read/audikw_1.mtx, as the reference's -m) instead of "
                                              "a synthetic workload; symmetric files of >= 45056 rows get symmetric pair storage (--sym-pairs auto)")
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS),
                    help="default: audikw_1-like at N = 1 (BASELINE config 2) and for --scaling weak, rmat-24 (config 5) for N > 1")
    ap.add_argument("--lds-doubles", type=int, default=0)
    ap.add_argument("--part-rows", type=int, default=0)
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--items-per-cu", type=int, default=0)
    ap.add_argument("--window-mode", type=int, default=0)
    ap.add_argument("--er-mode", type=int, default=0, help="residual form: 0/1 CSR segments, 2 panel form (cfg.er_mode)")
    ap.add_argument("--ell-nt", type=int, default=0, help="cfg.ell_nt (A/B: 1 the whole value stream past the caches, 2 plain loads as in rounds 1-3; default: plain loads for the end of an alternating walk only)")
    ap.add_argument("--balance", type=int, default=0, help="cfg.balance (symmetric pairs, A/B: 1 = partitions balanced on entries, 2 = on rows)")
    ap.add_argument("--graph-compress", type=int, default=0, help="cfg.graph_compress of the k-way partitioner (A/B: 1 on, 2 off, 3 on + refinement on the rows)")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the TIMED CPU legs (the parity check stays)")
    ap.add_argument("--no-overlap", action="store_true", help="N>1: exchange first, then multiply (no stream overlap)")
    ap.add_argument("--no-plain-arm", action="store_true",
                    help="N=1 with symmetric pair storage: skip the extra plain-storage measurement of the same matrix")
    ap.add_argument("--no-dropin-arm", action="store_true", help="N=1: skip the run through the reference-named calls")
    ap.add_argument("--ell-alternate", type=int, default=0,
                    help="cfg.ell_alternate: 0 = automatic (successive multiplies walk streams of 256 MB - 8 GB in alternating directions), 1 = always, 2 = never")
    ap.add_argument("--no-walk-arm", action="store_true", help="skip the side arm that times the same plan with every launch walking first to last")
    ap.add_argument("--no-live-pmc", action="store_true",
                    help="skip the two rocprofv3 --pmc child runs that measure roofline.traffic live (the table entry of the layout, else format bytes, is quoted)")
    ap.add_argument("--no-refill-arm", action="store_true",
                    help="N=1: build the plan without slot maps and skip the timing of the numeric phase on the device (ehyb_plan_set_values)")
    ap.add_argument("--no-scaling-anchor", action="store_true", help="N=1: skip the one-GPU run of the N>1 default workload")
    ap.add_argument("--vendor-baseline", action="store_true", help="N=1: also time rocSPARSE CSR SpMV on the same matrix (opt-in)")
    ap.add_argument("--sym-pairs", default="auto", choices=["auto", "on", "off"],
                    help="symmetric pair storage (cfg.sym_pairs): auto = on for the symmetric workloads")
    ap.add_argument("--scaling", default="strong", choices=["weak", "strong"],
                    help="N>1: strong = ONE matrix sharded by rows (default workload rmat-24, config 5); "
                         "weak = the N=1 matrix once per GPU, rank-local build, halo exchange (fem3d workloads)")
    ap.add_argument("--exchange", default=None, choices=["halo", "cover", "allgather"],
                    help="N>1 strong: halo = the x entries a rank's rows reference, hottest first, in chunks; cover = per pair of ranks only the "
                         "hub columns of the block travel as x, the rest of the block is handed to the columns' owner, who ships one partial "
                         "sum per row (a vertex cover of the block: 2.2-2.7 x fewer doubles on R-MAT 2^24; panel-form plans = the R-MAT workloads); "
                         "allgather = every x segment to everyone, padded to equal length.  Default: cover for the R-MAT workloads, halo otherwise")
    ap.add_argument("--chunks", type=int, default=2,
                    help="N>1 halo: exchange steps per multiply -- every owner's ghost columns, hottest first, are cut into this many "
                         "chunks; the panels of chunk k are multiplied while chunk k+1 is on the wire")
    ap.add_argument("--chunk-shares", default="", help="N>1 halo: share of every owner's ghost columns per chunk, e.g. 0.3,0.7 (default: 0.25,0.75 for two chunks, else equal)")
    ap.add_argument("--exchange-mode", default="a2a", choices=["a2a", "p2p"],
                    help="N>1 halo: a2a = one all_to_all_single per exchange step; p2p = grouped isend/irecv pairs (explicit, never a fallback)")
    ap.add_argument("--step", default="auto", choices=["auto", "c", "python"],
                    help="N>1: who issues the exchange -- c = libehyb.so's own RCCL communicator, ONE C call per multiply (ehyb_halo_spmv / "
                         "ehyb_gather_spmv); python = torch.distributed collectives issued from Python, one C call per part (the A/B arm, and "
                         "the only way over gloo); auto = c on the nccl backend")
    ap.add_argument("--no-single-gpu-anchor", action="store_true",
                    help="N>1 strong: skip the run of the same matrix on rank 0's GPU alone (the N = 1 point inside the N > 1 line)")
    ap.add_argument("--no-tune", action="store_true", help="N=1: skip ehyb_plan_tune (the item -> workgroup map stays the built-in one)")
    ap.add_argument("--verbose", action="store_true")
    args = ap.parse_args()
    args.host_us_per_step = 0.0

    world_env = os.environ.get("WORLD_SIZE")
    if args.gpus > 1 and world_env is None:
        self_launch(args)  # does not return
    world = int(world_env or "1")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node equal to --gpus")
    if args.workload is None:
        args.workload = "audikw_1-like" if (world == 1 or args.scaling == "weak") else "rmat-24"

    import numpy as np
    import torch

    import ehyb_spmv_gpu_amd as E

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the EHYB multiply has no CPU fallback")
    # EHYB_BENCH_ONE_DEVICE=1 + EHYB_BENCH_BACKEND=gloo: all ranks on cuda:0 over gloo -- a functional
    # test of the N > 1 path on a one-GPU box (not a measurement; RCCL refuses shared devices).
    one_device = os.environ.get("EHYB_BENCH_ONE_DEVICE") == "1"
    backend = os.environ.get("EHYB_BENCH_BACKEND", "nccl")
    if one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    def log(*a):
        if rank == 0:
            print(*a, file=sys.stderr, flush=True)

    kw = {}
    for k, v in (("lds_doubles", args.lds_doubles), ("part_rows", args.part_rows), ("threads", args.threads),
                 ("items_per_cu", args.items_per_cu), ("window_mode", args.window_mode), ("er_mode", args.er_mode),
                 ("ell_alternate", args.ell_alternate), ("graph_compress", args.graph_compress), ("balance", args.balance), ("ell_nt", args.ell_nt)):
        if v:
            kw[k] = v
    if world > 1 and os.environ.get("OMP_NUM_THREADS") == "1":
        # torch.distributed.run pins every rank to one OpenMP thread; the host pre-step (partitioner,
        # layout builder) of each rank gets its share of the CPUs the job owns instead (affinity mask and
        # cgroup quota -- a container may see many more hardware threads than it may use)
        kw["host_threads"] = max(1, owned_cpus() // world)
    gen, gargs, desc = WORKLOADS[args.workload]
    # Symmetric pair storage for matrices that are symmetric (the reference reads such files with
    # matrixRead_sym, solver_test.c:127-265, and knows it too): an in-partition pair is stored once.
    sym = args.sym_pairs == "on" or (args.sym_pairs == "auto" and symmetric_storage_pays(gen, gargs))
    file_matrix = None
    if args.mtx:
        if world > 1:
            raise SystemExit("bench.py: --mtx is an N = 1 option")
        # a real file (the reference's `-m name` = ./read/name.mtx): read once to learn its size and symmetry -- the
        # storage, and with it the partition sizing, follow from the banner as in solver_test (solver_test.c:348-354)
        t0 = time.time()
        probe = E.Matrix.read_mtx(args.mtx)
        sym = args.sym_pairs == "on" or (args.sym_pairs == "auto" and probe.symmetric and probe.n >= SYM_MIN_ROWS)
        file_matrix = (probe.symmetric, probe.n, probe.nnz)
        probe.free()
        gen, gargs = "file", (args.mtx,)
        args.workload = os.path.splitext(os.path.basename(args.mtx))[0]
        desc = f"{args.mtx} ({'symmetric' if file_matrix[0] else 'general'} Matrix Market file, {file_matrix[1]} rows, {file_matrix[2]} entries expanded)"
        args.no_plain_arm = args.no_dropin_arm = True   # those arms rebuild the matrix from its generator
        print(f"[bench] {desc}: read in {time.time() - t0:.1f}s", file=sys.stderr, flush=True)

    if world > 1:
        if sym:
            kw["sym_pairs"] = 1
        cfg = E.make_config(partitioner=partitioner_for(E, gen), verbose=1 if (args.verbose and rank == 0) else 0, **kw)
        if args.scaling == "weak":
            if gen != "fem3d":
                raise SystemExit("bench.py: --scaling weak stacks fem3d grids; use --scaling strong for " + args.workload)
            run_weak(args, E, torch, dist, rank, world, dev, cfg, log, stage_on_cpu=backend != "nccl")
        else:
            run_strong(args, E, torch, dist, rank, world, dev, cfg, log, stage_on_cpu=backend != "nccl")
        dist.destroy_process_group()
        return

    # ================================================================== N = 1
    from oracle import oracle as O

    if sym:
        kw["sym_pairs"] = 1
    cfg = E.make_config(partitioner=partitioner_for(E, gen), verbose=1 if args.verbose else 0, value_map=0 if args.no_refill_arm else 1, **kw)
    t0 = time.time()
    m = E.Matrix.read_mtx(args.mtx, cfg) if args.mtx else E.Matrix.generate(gen, *gargs, cfg=cfg)
    n, nnz = m.n, m.nnz
    log(f"[bench] {'read' if args.mtx else 'generated'} {args.workload}: n={n} nnz={nnz} in {time.time() - t0:.1f}s")

    # ---- CPU baseline on the un-permuted matrix: the oracle, timed; its y is also the parity checker
    cpu_baseline = None
    x = E.x_glibc(n)
    if args.no_cpu_baseline:
        y_cpu = O.spmv_coo(n, m.I, m.J, m.V, x)  # checker only, untimed
    else:
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
        log(f"[bench] cpu baseline: {cpu_baseline['value']} GFLOP/s on {cores} threads")
    scale = O.abs_rowsum(n, m.I, m.J, m.V, x)

    # ---- host pre-step: partition + permute (matrixReorder), then the plan
    t0 = time.time()
    m.reorder(cfg)
    t_reorder = time.time() - t0
    log(f"[bench] reorder (partition into {m.c.nParts} parts) {t_reorder:.3f}s")
    perm = m.reorder_list.copy()
    x_d = torch.from_numpy(E.vector_reorder(x, perm)).to(dev)   # (first device call of the process: the HIP context is up before the plan is timed)
    y_d = torch.zeros(n, dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    t0 = time.time()
    plan = E.Plan(m, cfg)
    t_plan = time.time() - t0
    st = plan.stats
    pre_step = {"reorder_s": round(t_reorder, 3), "plan_build_and_upload_s": round(t_plan, 3), "host_threads": int(E.host_threads()),
                "what": "ehyb_matrix_reorder (partition + permute), then ehyb_plan_create (layout on the host threads + copy to the device); tools/prestep_time.py has the phases"}
    log(f"[bench] plan built+uploaded in {t_plan:.3f}s: "
        f"ell {st['nnz_ell']} er {st['nnz_er']} pad {st['ell_padding']} items {st['n_items']} lds {st['lds_bytes']}B")
    stream = torch.cuda.current_stream().cuda_stream
    xp, yp = x_d.data_ptr(), y_d.data_ptr()
    tuned = None
    if not args.no_tune:
        # before the warm-up, outside the timed region: the item -> workgroup map for THIS device (ehyb_plan_tune)
        torch.cuda.synchronize()
        t0 = time.time()
        span0, span1 = plan.tune(xp, yp)
        tuned = {"api": "ehyb_plan_tune: the heaviest work items on the XCDs measured fastest (stamped launches before the warm-up)",
                 "stamped_launch_span_us_before": round(span0, 2), "stamped_launch_span_us_after": round(span1, 2), "seconds": round(time.time() - t0, 3)}
        if span0 == 0.0:
            tuned = None    # nothing to tune: no ELL launch, or more than one round of workgroups
        log(f"[bench] item map tuned for this device: stamped launch {span0:.1f} -> {span1:.1f} us")

    def step():
        plan.spmv(xp, yp, stream)

    elapsed = timed_steps(step, args, torch, dist, 1, dev)

    # ---- parity of what was just timed against the CPU oracle
    y = E.vector_recover(y_d.cpu().numpy(), perm)
    bad, worst = O.check_tolerance(y, y_cpu, scale)
    parity = {"rows_over_1e-12": bad, "worst_rel": worst}
    log(f"[bench] parity vs CPU oracle: {bad} rows over 1e-12, worst {worst:.3e}")
    if bad:
        raise SystemExit("bench.py: GPU result differs from the CPU oracle; refusing to report a number")

    # ---- per-kernel timing with HIP events on the launch stream; roofline of the dominant kernel
    r = plan.bench(xp, yp, stream, warmup=5, iters=min(args.steps, 200))
    ell_ms, er_ms = r["ms_ell_avg"], r["ms_er_avg"]
    inline = st["er_inline"] > 0  # the ELL launch also multiplies the (tiny) residual: one launch per SpMV
    empty = st["nnz_er"] == 0
    if inline or empty:
        er_ms = 0.0  # no residual launch: the interval between the two events is event overhead
    step_ms = r["ms_total"] / min(args.steps, 200)   # the graph-replayed loop between two HIP events on the launch stream
    kname = dominant_kernel(st, ell_ms, er_ms)
    traffic = pmc_traffic(args.workload, st["sym_pairs"] > 0, kname, st)
    traffic_table, live_detail = traffic, None
    if not args.no_live_pmc:
        # the counters taken by THIS run (child processes under rocprofv3, after the timed loop: nothing of it is inside `value`)
        live, live_detail = side_arm_pair("live PMC traffic", lambda: live_pmc_traffic(args.workload, st["sym_pairs"] > 0, kname, st, log, mtx=args.mtx), log)
        if live:
            traffic = live
        else:
            log(f"[bench] no live PMC measurement ({live_detail}): " + ("the table entry of this layout is quoted" if traffic else "format bytes are quoted"))
    basis = ("rocprofv3 PMC bytes per launch measured by this run (tools/pmc_run.py under --pmc FETCH_SIZE and --pmc WRITE_SIZE, "
             "separate passes, FETCH_SIZE calibrated on a 1 GiB streaming read in the same process)" if (traffic and traffic is not traffic_table) else
             "rocprofv3 PMC bytes per launch of this workload, storage and kernel (profiles/pmc_traffic.json)"
             if traffic else "format bytes per launch (what this layout makes the kernel move; no PMC measurement of exactly this layout on file)")
    roofline = roofline_block(st, ell_ms, er_ms, step_ms, traffic, basis)
    roofline["traffic_table"] = traffic_table
    roofline["traffic_live"] = live_detail if isinstance(live_detail, dict) else None
    achieved = roofline["achieved"]
    # the second ceiling SURVEY 8d asks for: a streaming read of 4 GiB measured on this box in this run
    try:
        import ctypes as C

        from ehyb_spmv_gpu_amd import _lib

        bw = C.c_double(0)
        if _lib.load().ehyb_measure_read_bw(C.c_size_t(1 << 32), 5, C.byref(bw)) == 0 and bw.value > 0:
            roofline["measured_read_ceiling_GBps"] = round(bw.value, 1)
            roofline["frac_of_measured_ceiling"] = round(achieved / bw.value, 4)
    except Exception:  # the probe is a convenience: never let it cost the bench line
        pass
    if st["sym_pairs"] > 0:
        roofline["note"] = ("symmetric pair storage: %d of the %d entries are in-partition pairs a_ij == a_ji stored once "
                            "(one value read, two FMAs, the mirror product added in LDS): frac counts the bytes really "
                            "moved; alg_frac prices every entry at 12 B (SURVEY 8d) and may exceed 1" % (2 * st["sym_pairs"], st["nnz"]))
    # ---- numeric phase of the build on the device: the plan's own values handed over again, as a host
    # array and as a device array, then the multiply checked once more.  After the timed loop: it cannot touch `value`.
    refill = None
    if not args.no_refill_arm:
        def refill_case():
            V = np.ascontiguousarray(m.V, dtype=np.float64)
            t0 = time.perf_counter()
            plan.set_values(V)                       # first call: uploads the slot maps too
            t_first = time.perf_counter() - t0
            t0 = time.perf_counter()
            plan.set_values(V)
            t_host = time.perf_counter() - t0
            v_d = torch.from_numpy(V).to(dev)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            plan.set_values((v_d.data_ptr(), len(V)), stream=stream)
            torch.cuda.synchronize()
            t_dev = time.perf_counter() - t0
            y_d.zero_()
            plan.spmv(xp, yp, stream)
            torch.cuda.synchronize()
            b2, w2 = O.check_tolerance(E.vector_recover(y_d.cpu().numpy(), perm), y_cpu, scale)
            if b2:
                raise SystemExit(f"refilled plan differs from the CPU oracle in {b2} rows")
            return {"api": "ehyb_plan_set_values (csrc/ehyb_fill.hip): new values on the plan's pattern gathered into the value streams on the GPU",
                    "ms_host_values": round(t_host * 1e3, 2), "ms_host_values_first_call": round(t_first * 1e3, 2),
                    "ms_device_values": round(t_dev * 1e3, 3), "slots": st["size_block_ell"] + st["er_inline"] + (0 if inline else st["nnz_er"]),
                    "parity_after": {"rows_over_1e-12": b2, "worst_rel": w2}}
        refill = side_arm("numeric-phase arm", refill_case, log)
        log(f"[bench] numeric phase on the device: {refill}")
    # ---- for the record: the same matrix and permutation with every launch walking its streams FIRST TO LAST (cfg.ell_alternate = 2).
    # `value` is measured with the library's default, where successive multiplies of a plan alternate the direction and a launch
    # starts with what the one before it left in the 256 MB Infinity Cache.
    walk = None
    if not args.no_walk_arm and args.ell_alternate == 0:
        def walk_case():
            cfg2 = E.make_config(partitioner=partitioner_for(E, gen), value_map=0, ell_alternate=2, **kw)
            p2 = E.Plan(m, cfg2)
            try:
                if not args.no_tune:
                    p2.tune(xp, yp)
                r2 = p2.bench(xp, yp, stream, warmup=args.warmup, iters=args.steps, per_kernel=False)
            finally:
                p2.destroy()
            t2 = r2["ms_total"] / args.steps
            return {"first_to_last_GFLOPs": round(2.0 * nnz / t2 / 1e6, 2), "first_to_last_ms_per_step": round(t2, 5),
                    "alternating_is_the_default": "cfg.ell_alternate = 0: successive multiplies of a plan walk streams of 256 MB - 8 GB in alternating directions"}
        walk = side_arm("first-to-last arm", walk_case, log)
        if walk and walk.get("first_to_last_ms_per_step"):
            add_first_to_last(roofline, walk["first_to_last_ms_per_step"], step_ms)
        log(f"[bench] every launch first to last (ell_alternate = 2): {walk}")
    plan.destroy()
    del x_d, y_d
    m.free()

    # ---- the same matrix with plain storage (every entry stored, as the reference does), for the record
    plain = None
    if st["sym_pairs"] > 0 and not args.no_plain_arm:
        t0 = time.time()
        plain = side_arm("plain-storage arm", lambda: one_gpu_case(E, O, np, args.workload, False, kw, args.steps, args.warmup, log,
                                                                   y_cpu=y_cpu, scale=scale, x=x), log)
        plain["note"] = "every entry stored (bench.py --sym-pairs off); same run, same GPU"
        log(f"[bench] plain-storage arm: {plain.get('value')} GFLOP/s ({time.time() - t0:.1f}s incl. its own pre-step)")

    # ---- the same matrix through the reference-named calls with the reference driver's sizing
    dropin = None
    if args.workload in REFERENCE_SIZING and not args.no_dropin_arm:
        t0 = time.time()
        dropin = side_arm("drop-in path", lambda: dropin_case(E, O, np, args.workload, x, y_cpu, scale, args.steps, log), log)
        log(f"[bench] drop-in path: {dropin.get('value')} GFLOP/s with {dropin.get('nParts_used')} partitions ({time.time() - t0:.1f}s)")

    # ---- vendor library on the same matrix (opt-in; not part of the product path)
    vendor = None
    if args.vendor_baseline:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import compare_rocsparse as R

        vendor = side_arm("vendor baseline", lambda: R.rocsparse_arm(E, O, np, gen, gargs, x, y_cpu, scale, iters=min(args.steps, 100)), log)
        log(f"[bench] rocSPARSE CSR: {vendor}")

    # ---- N = 1 point of the strong-scaling curve: the N > 1 default workload on this GPU
    anchor = None
    if args.workload == "audikw_1-like" and not args.no_scaling_anchor:
        t0 = time.time()
        del y_cpu, scale
        anchor = side_arm("scaling anchor", lambda: one_gpu_case(E, O, np, "rmat-24", False, {}, min(args.steps, 50), min(args.warmup, 5), log), log)
        anchor["workload"] = "rmat-24"
        anchor["note"] = "BASELINE config 5's matrix on ONE GPU: the N = 1 point for `bench.py --gpus N` (strong scaling, same matrix)"
        log(f"[bench] scaling anchor rmat-24 on one GPU: {anchor.get('value')} GFLOP/s, {anchor.get('ms_per_step')} ms ({time.time() - t0:.1f}s)")

    ms_per_step = elapsed / args.steps * 1e3
    out = {
        "metric": "fp64 SpMV GFLOP/s (2*nnz/t_iter), EHYB on MI355X",
        "value": round(2.0 * nnz * args.steps / elapsed / 1e9, 2), "unit": "GFLOP/s", "n_gpus": 1, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms_per_step, 5), "higher_is_better": True,
        "scaling": "none", "vs_baseline": None, "dtype": "f64", "data": ("file: " if args.mtx else "synthetic: ") + desc,
        "config": {"workload": args.workload, "rows": n, "nnz": nnz, "parts": st["n_parts"],
                   "lds_doubles": int(cfg.lds_doubles), "part_rows": int(cfg.part_rows), "threads": int(cfg.threads),
                   "window_mode": "halo" if cfg.window_mode != 1 else "reference",
                   "nnz_ell": st["nnz_ell"], "nnz_er": st["nnz_er"], "ell_padding": st["ell_padding"],
                   "sym_pairs": st["sym_pairs"], "stored_values": st["size_block_ell"],
                   "alg_bytes_per_spmv": st["bytes_alg"], "exchange": "none"},
        "alg_GBps": round((12 * nnz + 4 * (n + 1) + 16 * n) / (elapsed / args.steps) / 1e9, 1),
        "roofline": roofline, "cpu_baseline": cpu_baseline, "parity": parity, "walk": walk,
    }
    out["pre_step"] = pre_step
    if tuned:
        out["tuned_item_map"] = tuned
    if refill:
        out["numeric_refill"] = refill
    if plain:
        out["plain_storage"] = plain
    if dropin:
        out["dropin_path"] = dropin
    if vendor:
        out["vendor_baseline"] = vendor
    if anchor:
        out["scaling_anchor"] = anchor
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
