"""Where the host pre-step of the bench matrix spends its time: reorder (adjacency / partition / row order / permute) and the plan
(pair orientation / window pass / fill; with --upload the copy to the device as well), the library's own laps (cfg.verbose = 2),
best of --reps builds.  One JSON line.
    python tools/prestep_time.py [--workload audikw_1-like] [--reps 4] [--upload] [--col-map 2] [--threads 8]"""
import argparse
import json
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

WORKLOADS = {
    "audikw_1-like": ("fem3d", (943695, 3, 68, 68, 13500, 1, 1), True),
    "audikw_1-graded": ("fem3d_graded", (943695, 3, 68, 68, 100000, 705000, 1, 1), True),
    "audikw_1-plain": ("fem3d", (943695, 3, 68, 68, 13500, 1, 1), False),
    "small": ("fem3d", (120000, 3, 35, 35, 13500, 1, 1), True),
}


def child(a):
    import ehyb_spmv_gpu_amd as E
    gen, gargs, sym = WORKLOADS[a.workload]
    kw = dict(verbose=2, partitioner=E.EHYB_PART_AUTO, value_map=1, col_map=a.col_map, host_threads=a.threads)
    if sym:
        kw["sym_pairs"] = 1
    cfg = E.make_config(**kw)
    t = time.time()
    m = E.Matrix.generate(gen, *gargs, cfg=cfg)
    print("TOTAL generate %.1f ms" % ((time.time() - t) * 1e3), flush=True)
    t = time.time()
    m.reorder(cfg)
    print("TOTAL reorder %.1f ms" % ((time.time() - t) * 1e3), flush=True)
    for _ in range(a.reps):
        t = time.time()
        p = E.Plan(m, cfg=cfg, upload=a.upload)
        print("TOTAL plan%s %.1f ms" % ("+upload" if a.upload else "", (time.time() - t) * 1e3), flush=True)
        p.destroy()
    print("THREADS %d" % E.host_threads(), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="audikw_1-like", choices=sorted(WORKLOADS))
    ap.add_argument("--reps", type=int, default=4)
    ap.add_argument("--upload", action="store_true")
    ap.add_argument("--col-map", type=int, default=0)
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--child", action="store_true")
    a = ap.parse_args()
    if a.child:
        return child(a)
    # (the library prints its laps with printf: read them from a child's stdout)
    out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"] + sys.argv[1:], capture_output=True, text=True)
    if out.returncode:
        sys.stderr.write(out.stdout + out.stderr)
        raise SystemExit(out.returncode)
    best, order, threads, first_plan = {}, [], 0, None
    pat = re.compile(r"\s*(layout: .*?|TOTAL [\w+]+|permute time is|partition time is|row order time is|adjacency time is|k-way partition time is)\s+([\d.]+) (ms|us)")
    for line in out.stdout.splitlines():
        if line.startswith("THREADS"):
            threads = int(line.split()[1])
        sub = re.match(r"\s*permute: setup ([\d.]+) prefault ([\d.]+) gather ([\d.]+) free ([\d.]+) s", line)
        if sub:
            for k, v in zip(("permute: setup", "permute: fresh pages", "permute: gather", "permute: free"), sub.groups()):
                k = "reorder: " + k
                if k not in best:
                    order.append(k)
                best[k] = min(best.get(k, 1e18), float(v) * 1e3)
            continue
        sub = re.match(r"partition: n=\d+ parts=\d+ levels=\d+ cut=\d+\s+coarsen ([\d.]+)s init ([\d.]+)s refine ([\d.]+)s", line)
        if sub:
            for k, v in zip(("k-way: coarsen", "k-way: initial partition", "k-way: refine"), sub.groups()):
                if k not in best:
                    order.append(k)
                best[k] = min(best.get(k, 1e18), float(v) * 1e3)
            continue
        mm = pat.match(line)
        if not mm:
            continue
        if mm.group(1).startswith("TOTAL plan") and first_plan is None:
            first_plan = float(mm.group(2))   # the build right behind the reorder (what a caller gets; the best of --reps is the builder alone)
        k = mm.group(1).strip().replace("layout: ", "plan: ").replace(" time is", "").replace("TOTAL ", "total ")
        v = float(mm.group(2)) / (1000 if mm.group(3) == "us" else 1)
        if k not in best:
            order.append(k)
        best[k] = min(best.get(k, 1e18), v)
    print(json.dumps({"workload": a.workload, "host_threads": threads, "col_map": a.col_map, "reps": a.reps,
                      "first_plan_build_ms": first_plan, "ms": {k: round(best[k], 1) for k in order if best[k] >= 0.05}}))


if __name__ == "__main__":
    main()
