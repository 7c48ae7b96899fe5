"""Row-sharded EHYB SpMV across the GPUs of one node: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI) for the exchange of x -- SURVEY.md 8e.

The reference is single-GPU; this is the new multi-GPU layer the north star asks for.  The
matrix is partitioned in two levels by ehyb_matrix_reorder (cfg.n_top = world size): rank r owns
the rows (and the matching x / y segment) of top-level block r.  Its plan multiplies the block's
rows against the full-length x:
    phase 1 (ELL)       needs only the rank's own x segment (window columns are local),
    phase 2 (residual)  needs the other ranks' segments.
One iteration = exchange of the x segments (all-gatherv) + local multiply; `overlap=True` runs
phase 1 on the compute stream while the exchange is in flight on a side stream.

torch is plumbing here: device buffers, streams and the collective.  The kernels are libehyb.so's.
"""
import ctypes as C

import numpy as np

from . import host as H


def row_cuts(matrix, cfg, world):
    """First row of every rank's block, world+1 entries (whole partitions per rank)."""
    blocks = (C.c_int * (world + 1))()
    H._check(H._lib.load().ehyb_top_boundary(C.byref(matrix.c), C.byref(cfg) if cfg else None, world, blocks),
             "ehyb_top_boundary")
    pb = matrix.part_boundary
    return [int(pb[blocks[b]]) for b in range(world + 1)]


def exchange_segments(x_full, cuts, rank, group=None):
    """All-gatherv: every rank publishes x_full[cuts[rank]:cuts[rank+1]], all end with the full
    vector.  Equal segments -> one all_gather_into_tensor; ragged -> list form (RCCL: grouped
    broadcasts) or, on backends without ragged support (gloo), one broadcast per segment."""
    import torch.distributed as dist

    world = len(cuts) - 1
    if world == 1:
        return
    sizes = {cuts[b + 1] - cuts[b] for b in range(world)}
    mine = x_full[cuts[rank]:cuts[rank + 1]]
    if len(sizes) == 1:
        dist.all_gather_into_tensor(x_full[cuts[0]:cuts[world]], mine, group=group)
        return
    if dist.get_backend(group) == "nccl":
        dist.all_gather([x_full[cuts[b]:cuts[b + 1]] for b in range(world)], mine, group=group)
        return
    for b in range(world):
        dist.broadcast(x_full[cuts[b]:cuts[b + 1]], src=dist.get_global_rank(group, b) if group else b, group=group)


class ShardedSpmv:
    """One rank's share of the row-sharded multiply on its GPU."""

    def __init__(self, matrix, cfg, rank, world, device, group=None, overlap=True):
        import torch

        self.torch = torch
        self.rank, self.world, self.group = rank, world, group
        self.n = matrix.n
        self.cuts = row_cuts(matrix, cfg, world) if world > 1 else [0, matrix.n]
        self.r0, self.r1 = self.cuts[rank], self.cuts[rank + 1]
        self.plan = H.Plan(matrix, cfg, rows=(self.r0, self.r1))
        self.device = device
        self.x = torch.zeros(self.n, dtype=torch.float64, device=device)
        self.y = torch.zeros(self.n, dtype=torch.float64, device=device)
        self.overlap = overlap and world > 1
        self.comm_stream = torch.cuda.Stream(device=device) if self.overlap else None

    def set_x(self, x_perm):
        self.x.copy_(self.torch.from_numpy(np.ascontiguousarray(x_perm)))

    def step(self):
        """x segments -> everyone, then y[r0:r1] = A[r0:r1, :] x."""
        torch = self.torch
        cur = torch.cuda.current_stream()
        if self.world == 1:
            self.plan.spmv(self.x.data_ptr(), self.y.data_ptr(), cur.cuda_stream)
            return
        if not self.overlap:
            exchange_segments(self.x, self.cuts, self.rank, self.group)
            self.plan.spmv(self.x.data_ptr(), self.y.data_ptr(), cur.cuda_stream)
            return
        self.comm_stream.wait_stream(cur)
        with torch.cuda.stream(self.comm_stream):
            exchange_segments(self.x, self.cuts, self.rank, self.group)
        self.plan.spmv(self.x.data_ptr(), self.y.data_ptr(), cur.cuda_stream, phase=1)  # local columns only
        cur.wait_stream(self.comm_stream)
        self.plan.spmv(self.x.data_ptr(), self.y.data_ptr(), cur.cuda_stream, phase=2)  # remote columns

    def local_y(self):
        return self.y[self.r0:self.r1]
