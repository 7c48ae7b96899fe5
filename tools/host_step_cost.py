#!/usr/bin/env python3
"""Host microseconds per call of what one multi-GPU step issues, measured on ONE GPU: the torch.distributed collective (RCCL
communicator with a single rank: the c10d + ncclGroup + launch path is the same, the wire is not), `ehyb_step_pack`,
`ehyb_step_part`.  host_us_per_step of an N-GPU run ~ pack + (K + 1) parts + K collectives.

usage: python tools/host_step_cost.py [--doubles 2000000] [--reps 300]
"""
import argparse
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--doubles", type=int, default=2_000_000)
    ap.add_argument("--reps", type=int, default=300)
    args = ap.parse_args()
    import numpy as np
    import torch
    import torch.distributed as dist

    import ehyb_spmv_gpu_amd as E
    from ehyb_spmv_gpu_amd import _lib

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29591")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    lib = _lib.load()
    n = args.doubles
    send = torch.ones(n, dtype=torch.float64, device=dev)
    recv = torch.empty(n, dtype=torch.float64, device=dev)
    comm = torch.cuda.Stream(device=dev)
    cur = torch.cuda.current_stream()
    out = {}

    def timed(name, fn, batch=8):
        """Host time to ISSUE one call: batches of `batch` calls on an idle device (a queue that runs full -- HIP's, or RCCL's work
        FIFO -- would make the host wait for the device and the figure the device's), and the time per call incl. the device."""
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        t_issue = 0.0
        t0_all = time.perf_counter()
        done = 0
        while done < args.reps:
            t0 = time.perf_counter()
            for _ in range(batch):
                fn()
            t_issue += time.perf_counter() - t0
            torch.cuda.synchronize()
            done += batch
        t_all = time.perf_counter() - t0_all
        out[name] = {"host_us_per_call": round(t_issue / done * 1e6, 1), "us_per_call_incl_device_and_sync": round(t_all / done * 1e6, 1)}

    def a2a():
        with torch.cuda.stream(comm):
            dist.all_to_all_single(recv, send, [n], [n])

    timed("all_to_all_single (nccl, 1 rank, %d doubles, on a side stream)" % n, a2a)
    idx = torch.arange(n, dtype=torch.int32, device=dev)
    timed("ehyb_step_pack", lambda: lib.ehyb_step_pack(C.c_void_p(send.data_ptr()), C.c_void_p(idx.data_ptr()), C.c_void_p(recv.data_ptr()), n,
                                                       C.c_void_p(cur.cuda_stream), C.c_void_p(comm.cuda_stream)))
    # a small panel-form plan with three column segments: the parts of a step
    cfg = E.make_config(er_mode=2, fuse_er=2, n_top=2)
    m = E.Matrix.generate("rmat", 18, 1 << 21, 1, cfg=cfg)
    m.reorder(cfg)
    nn = m.n
    segs = np.array([0, nn // 2 & ~1, (3 * nn // 4) & ~1, nn], dtype=np.int32)
    plan = E.Plan(m, cfg, col_segs=segs)
    x = torch.ones(nn, dtype=torch.float64, device=dev)
    y = torch.zeros(nn, dtype=torch.float64, device=dev)

    def parts():
        lib.ehyb_step_part(plan.h, C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), C.c_void_p(cur.cuda_stream), C.c_void_p(comm.cuda_stream), 0, 0, 1, 1)
        lib.ehyb_step_part(plan.h, C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), C.c_void_p(cur.cuda_stream), C.c_void_p(comm.cuda_stream), 1, 1, 2, 0)
        lib.ehyb_step_part(plan.h, C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), C.c_void_p(cur.cuda_stream), C.c_void_p(comm.cuda_stream), 1, 2, 3, 2)

    timed("three ehyb_step_part calls (own columns, chunk 0, chunk 1 + closing pass; R-MAT 2^18)", parts)
    # ---- the whole step as ONE C call over libehyb.so's own RCCL communicator (csrc/ehyb_comm.hip): a rank-local R-MAT whose
    # ghost columns are 60 % of its own columns, sent to itself in two chunks (0.25 / 0.75, bench.py's default) -- the ncclGroup +
    # launch path of a real step without the wire
    from ehyb_spmv_gpu_amd import dist as D

    cfg_g = E.make_config(partitioner=E.EHYB_PART_DEGREE)
    cfg_p = E.make_config(partitioner=E.EHYB_PART_DEGREE, er_mode=2)
    mm = E.Matrix.generate("rmat", 18, 1 << 21, 1, cfg=cfg_g)
    I, J, V, n_r = mm.I.copy(), mm.J.copy(), mm.V.copy(), mm.n
    mm.free()
    L = D.RankLocalMatrix(I, J, V, [0, n_r], 0, cfg_p, chunks=2, chunk_shares=[0.25, 0.75], loopback=0.6)
    c = D.Comm(D.Comm.unique_id(), 0, 1)
    sh_c = D.HaloSpmv(L, dev, comm=c)
    sh_py = D.HaloSpmv(L, dev)                      # the same step issued part by part from Python (device copies as the collective)
    for sh in (sh_c, sh_py):
        sh.set_x_local(np.ones(n_r))
    timed("WHOLE STEP, one C call: ehyb_halo_spmv over RCCL (pack + 2 grouped send/recv exchanges + 3 parts + closing pass; R-MAT 2^18, %d ghost columns)" % L.n_ghost,
          sh_c.step)
    timed("WHOLE STEP issued from Python: ehyb_step_pack + 3 ehyb_step_part + 2 device copies in place of the collectives", sh_py.step)
    import json

    print(json.dumps(out, indent=1))
    c.destroy()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
