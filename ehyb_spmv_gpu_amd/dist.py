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
import os

import numpy as np

from . import host as H


def row_cuts(matrix, cfg, world):
    """First row of every rank's block, world+1 entries (whole partitions per rank): the top-level
    blocks of the two-level reorder (Matrix.reorder with cfg.n_top == world) when the matrix carries
    them, else runs of whole partitions with equal entry counts (ehyb_top_boundary)."""
    blocks = getattr(matrix, "block_first", None)
    if blocks is None or len(blocks) != world + 1 or blocks[-1] != matrix.c.nParts:
        blocks = (C.c_int * (world + 1))()
        H._check(H._lib.load().ehyb_top_boundary(C.byref(matrix.c), C.byref(cfg) if cfg else None, world, blocks),
                 "ehyb_top_boundary")
    pb = matrix.part_boundary
    return [int(pb[blocks[b]]) for b in range(world + 1)]


def balanced_row_cuts(rowptr, world):
    """Row blocks of (nearly) equal entry counts, one per GPU: world+1 first rows.  The top level of the
    two-level partition of SURVEY 8e for callers that shard the rows themselves (every GPU streams the
    same number of bytes)."""
    rowptr = np.asarray(rowptr, dtype=np.int64)
    n = len(rowptr) - 1
    cuts = [0]
    for b in range(1, world):
        c = int(np.searchsorted(rowptr, rowptr[n] * b // world, side="left"))
        cuts.append(min(max(c, cuts[-1] + 1), n - (world - b)))
    cuts.append(n)
    return cuts


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


# ======================================================================================
# Rank-local build with a halo exchange.
#
# ShardedSpmv above needs the whole matrix on every rank and moves the whole of x every
# iteration.  At scale neither is acceptable: a rank should hold its own rows only and receive
# only the x entries its rows reference.  The classes below do that:
#
#   RankLocalMatrix   rows [r0, r1) of the global matrix (global column labels) ->
#                       * the square diagonal block, partitioned and permuted by
#                         ehyb_matrix_reorder on this rank alone,
#                       * "ghost" columns: one slot per distinct remote column, ordered by owner
#                         rank and then by global label, appended behind the local columns
#                         (ehyb_matrix_append_ghosts),
#                       * the send lists: which of its own (permuted) x entries every peer wants.
#   HaloExchange      per iteration: gather the send entries into one buffer, one all_to_all
#                     (RCCL over xGMI; point-to-point pairs on gloo) straight into the ghost part
#                     of x = [local x | ghosts].
#   HaloSpmv          the rank's multiply: phase 1 (ELL, local columns) runs while the exchange is
#                     in flight on a side stream, phase 2 (residual = all ghost columns) after it.
#
# Exchange volume per rank = number of ghost slots, not the length of x: for a mesh-like matrix
# cut into slabs that is the two interface layers, a few per cent of the rank's rows.
def _copy_cfg(cfg, **kw):
    c = type(cfg).from_buffer_copy(cfg) if cfg is not None else H.make_config()
    for k, v in kw.items():
        setattr(c, k, v)
    return c


class RankLocalMatrix:
    def __init__(self, I, J, V, cuts, rank, cfg=None, symmetric=False, group=None, exchange="halo"):
        """I, J, V: this rank's rows in global labels, row-grouped (I ascending).  cuts: first row of
        every rank, world+1 entries.  Collective: every rank of `group` must call it.
        exchange = "halo": one ghost slot per distinct remote column, filled by an all_to_all of exactly
        those entries (HaloSpmv).  exchange = "allgather": the ghost columns are the places of the remote
        entries inside the buffer an all-gather of the (padded) x segments fills (GatherSpmv):
        x = [own segment, padded to seg_len | segment of rank 0 | ... | segment of rank world-1], every
        segment in its owner's plan order, so nothing has to be unpacked after the collective."""
        world = len(cuts) - 1
        self.rank, self.world, self.cuts, self.group = rank, world, [int(c) for c in cuts], group
        r0, r1 = self.cuts[rank], self.cuts[rank + 1]
        self.r0, self.r1, self.n_loc = r0, r1, r1 - r0
        I = np.asarray(I)
        J = np.asarray(J)
        V = np.asarray(V, dtype=np.float64)
        if len(I) and (I.min() < r0 or I.max() >= r1 or np.any(np.diff(I) < 0)):
            raise ValueError("RankLocalMatrix: rows must lie in [r0, r1) and be grouped in ascending order")
        own = (J >= r0) & (J < r1)
        # diagonal block: a masked row-grouped sequence is still row-grouped
        indptr = np.zeros(self.n_loc + 1, dtype=np.int64)
        np.cumsum(np.bincount(I[own] - r0, minlength=self.n_loc), out=indptr[1:])
        cfg1 = _copy_cfg(cfg, n_top=1)
        self.m = H.Matrix.from_csr(indptr, J[own] - r0, V[own], cfg1, symmetric=symmetric)
        self.m.reorder(cfg1)
        self.perm = self.m.reorder_list[:self.n_loc].copy()  # local row i (unpermuted) -> its place in the plan
        # ghost slots: distinct remote columns, ascending = grouped by owner
        off = ~own
        Jg = J[off]
        gcols = np.unique(Jg)
        owner = np.searchsorted(np.asarray(self.cuts), gcols, side="right") - 1
        self.n_ghost = len(gcols)
        self.ghost_cols = gcols
        self.recv_counts = np.bincount(owner, minlength=world).astype(np.int64)
        assert self.recv_counts[rank] == 0
        wants = [gcols[owner == p] for p in range(world)]
        # what the peers want from me, in their slot order
        if world > 1:
            import torch.distributed as dist

            everyone = [None] * world
            dist.all_gather_object(everyone, wants, group=group)
        else:
            everyone = [wants]
        asked = [np.asarray(everyone[q][rank], dtype=np.int64) for q in range(world)]
        for a in asked:
            if len(a) and (a.min() < r0 or a.max() >= r1):
                raise ValueError("RankLocalMatrix: a peer asked for a column this rank does not own")
        self.send_counts = np.array([len(a) for a in asked], dtype=np.int64)
        places = [self.perm[a - r0].astype(np.int64) for a in asked]   # where, in my plan order, what each peer wants sits
        self.send_idx = np.concatenate(places) if world > 1 else np.zeros(0, np.int64)
        self.exchange = exchange
        self.seg_len = max(self.cuts[b + 1] - self.cuts[b] for b in range(world))
        if exchange == "allgather" and world > 1:
            import torch.distributed as dist

            told = [None] * world
            dist.all_gather_object(told, places, group=group)
            # ghost column of remote entry k of owner p: behind my padded segment, inside p's segment of the gathered buffer
            slot = np.empty(self.n_ghost, dtype=np.int64)
            at = 0
            for p in range(world):
                pl = np.asarray(told[p][rank], dtype=np.int64)
                slot[at:at + len(pl)] = (self.seg_len - self.n_loc) + p * self.seg_len + pl
                at += len(pl)
            assert at == self.n_ghost
            self.n_ext = (self.seg_len - self.n_loc) + world * self.seg_len
            self.m.append_ghosts(self.n_ext, self.perm[I[off] - r0], slot[np.searchsorted(gcols, Jg)], V[off])
        else:
            # the coupling entries, rows in plan numbering, columns = ghost slots
            self.n_ext = self.n_ghost
            self.m.append_ghosts(self.n_ghost, self.perm[I[off] - r0], np.searchsorted(gcols, Jg), V[off])
        self.nnz = len(V)
        self.cfg_plan = _copy_cfg(cfg, n_top=2 if world > 1 else 1)

    def plan(self, upload=True):
        return H.Plan(self.m, self.cfg_plan, rows=(0, self.n_loc), upload=upload)

    def x_to_plan(self, x_local):
        """Local x segment (global label order) -> plan order."""
        return H.vector_reorder(x_local, self.perm)

    def y_from_plan(self, y_plan):
        return H.vector_recover(y_plan, self.perm)


class HaloExchange:
    """x_ext = [local x (plan order) | ghost slots]; run() refreshes the ghost slots from the peers."""

    def __init__(self, local, x_ext, group=None, stage_on_cpu=False):
        import torch
        import torch.distributed as dist

        self.torch, self.dist = torch, dist
        self.L, self.group, self.x_ext = local, group, x_ext
        dev = x_ext.device
        self.send_idx = torch.from_numpy(local.send_idx).to(dev)
        self.send_buf = torch.empty(len(local.send_idx), dtype=torch.float64, device=dev)
        self.ghosts = x_ext[local.n_loc:]
        self.send_counts = [int(c) for c in local.send_counts]
        self.recv_counts = [int(c) for c in local.recv_counts]
        # RCCL: one all_to_all_single with uneven splits; EHYB_HALO_P2P=1 forces grouped send/recv pairs
        self.a2a = local.world > 1 and dist.get_backend(group) == "nccl" and os.environ.get("EHYB_HALO_P2P") != "1"
        self.stage = stage_on_cpu and dev.type != "cpu"  # gloo cannot move GPU tensors point to point

    def pack(self):
        if len(self.send_idx):
            self.torch.index_select(self.x_ext[:self.L.n_loc], 0, self.send_idx, out=self.send_buf)

    def transfer(self):
        L, dist, torch = self.L, self.dist, self.torch
        if L.world == 1:
            return
        if self.a2a:
            try:
                dist.all_to_all_single(self.ghosts, self.send_buf, self.recv_counts, self.send_counts, group=self.group)
                return
            except RuntimeError as err:  # a backend without uneven splits: grouped send/recv pairs from here on
                import sys

                print(f"[ehyb] all_to_all_single failed ({err}); falling back to batched isend/irecv", file=sys.stderr, flush=True)
                self.a2a = False
        send, recv = (self.send_buf.cpu(), torch.empty(L.n_ghost, dtype=torch.float64)) if self.stage else (self.send_buf, self.ghosts)
        ops, so, ro = [], 0, 0
        for q in range(L.world):
            peer = dist.get_global_rank(self.group, q) if self.group else q
            if self.send_counts[q]:
                ops.append(dist.P2POp(dist.isend, send[so:so + self.send_counts[q]], peer, group=self.group))
            if self.recv_counts[q]:
                ops.append(dist.P2POp(dist.irecv, recv[ro:ro + self.recv_counts[q]], peer, group=self.group))
            so += self.send_counts[q]
            ro += self.recv_counts[q]
        for r in (dist.batch_isend_irecv(ops) if ops else []):
            r.wait()
        if self.stage:
            self.ghosts.copy_(recv)

    def run(self):
        self.pack()
        self.transfer()


class HaloSpmv:
    """One rank's multiply of a RankLocalMatrix on its GPU: y_loc = A[r0:r1, :] x."""

    def __init__(self, local, device, overlap=True, stage_on_cpu=False):
        import torch

        self.torch = torch
        self.L = local
        self.plan = local.plan()
        self.x = torch.zeros(local.n_loc + local.n_ghost, dtype=torch.float64, device=device)
        self.y = torch.zeros(local.n_loc, dtype=torch.float64, device=device)
        self.halo = HaloExchange(local, self.x, local.group, stage_on_cpu=stage_on_cpu)
        self.overlap = overlap and local.world > 1
        self.comm_stream = torch.cuda.Stream(device=device) if self.overlap else None

    def set_x_local(self, x_local):
        """x_local: this rank's x segment in global label order."""
        self.x[:self.L.n_loc].copy_(self.torch.from_numpy(self.L.x_to_plan(x_local)))

    def step(self):
        torch = self.torch
        cur = torch.cuda.current_stream()
        xp, yp = self.x.data_ptr(), self.y.data_ptr()
        if self.L.world == 1:
            self.plan.spmv(xp, yp, cur.cuda_stream)
            return
        self.halo.pack()
        if not self.overlap:
            self.halo.transfer()
            self.plan.spmv(xp, yp, cur.cuda_stream)
            return
        self.comm_stream.wait_stream(cur)
        with torch.cuda.stream(self.comm_stream):
            self.halo.transfer()
        self.plan.spmv(xp, yp, cur.cuda_stream, phase=1)  # local columns only
        cur.wait_stream(self.comm_stream)
        self.plan.spmv(xp, yp, cur.cuda_stream, phase=2)  # ghost columns

    def y_local(self):
        """This rank's y segment in global label order (host array)."""
        return self.L.y_from_plan(self.y.cpu().numpy())

    def time_local(self, iters):
        return _time_local(self, iters)


def _time_local(sh, iters):
    """Mean milliseconds of this rank's multiply alone (both phases back to back, no exchange)."""
    torch = sh.torch
    cur = torch.cuda.current_stream()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    xp, yp = sh.x.data_ptr(), sh.y.data_ptr()
    for _ in range(3):
        sh.plan.spmv(xp, yp, cur.cuda_stream)
    a.record(cur)
    for _ in range(iters):
        sh.plan.spmv(xp, yp, cur.cuda_stream)
    b.record(cur)
    b.synchronize()
    return a.elapsed_time(b) / iters


class GatherSpmv:
    """One rank's multiply of a RankLocalMatrix(exchange="allgather") on its GPU: per step ONE
    all_gather_into_tensor of the padded x segments (RCCL over xGMI: the all-gatherv of north_star with
    equal counts) straight into the part of x the residual phase reads, overlapped with the ELL phase,
    which needs local columns only.  No pack or unpack kernel on either side."""

    def __init__(self, local, device, overlap=True, stage_on_cpu=False):
        import torch
        import torch.distributed as dist

        assert local.exchange == "allgather" or local.world == 1
        self.torch, self.dist = torch, dist
        self.L = local
        self.plan = local.plan()
        self.seg_len = local.seg_len
        self.x = torch.zeros(local.n_loc + local.n_ext, dtype=torch.float64, device=device)
        self.y = torch.zeros(local.n_loc, dtype=torch.float64, device=device)
        self.mine = self.x[:self.seg_len]                                   # own segment, zero padded
        self.gathered = self.x[self.seg_len:] if local.world > 1 else None  # world x seg_len
        self.overlap = overlap and local.world > 1
        self.comm_stream = torch.cuda.Stream(device=device) if self.overlap else None
        self.stage = stage_on_cpu and device.type != "cpu"                  # gloo cannot move GPU tensors

    def set_x_local(self, x_local):
        self.x[:self.L.n_loc].copy_(self.torch.from_numpy(self.L.x_to_plan(x_local)))

    def exchange(self):
        if self.L.world == 1:
            return
        if not self.stage:
            self.dist.all_gather_into_tensor(self.gathered, self.mine, group=self.L.group)
            return
        mine = self.mine.cpu()
        parts = [self.torch.empty_like(mine) for _ in range(self.L.world)]
        self.dist.all_gather(parts, mine, group=self.L.group)
        self.gathered.copy_(self.torch.cat(parts))

    def step(self):
        torch = self.torch
        cur = torch.cuda.current_stream()
        xp, yp = self.x.data_ptr(), self.y.data_ptr()
        if self.L.world == 1:
            self.plan.spmv(xp, yp, cur.cuda_stream)
            return
        if not self.overlap:
            self.exchange()
            self.plan.spmv(xp, yp, cur.cuda_stream)
            return
        self.comm_stream.wait_stream(cur)
        with torch.cuda.stream(self.comm_stream):
            self.exchange()
        self.plan.spmv(xp, yp, cur.cuda_stream, phase=1)  # local columns only
        cur.wait_stream(self.comm_stream)
        self.plan.spmv(xp, yp, cur.cuda_stream, phase=2)  # columns inside the gathered segments

    def y_local(self):
        return self.L.y_from_plan(self.y.cpu().numpy())

    def time_local(self, iters):
        return _time_local(self, iters)


class HaloCG:
    """(Jacobi-)preconditioned conjugate gradients over the ranks of a HaloSpmv: the iterating caller
    of the multi-GPU path ("iterating x <- y requires every GPU to obtain the other
    segments" -- here the direction vector p is what travels, through the halo exchange inside
    HaloSpmv.step).  Every rank holds its rows of x, r, p, q in plan order on its GPU.  The vector work
    is done by the fused kernels of the single-GPU solver (csrc/ehyb_cg.hip, ehyb_cg_*_step: three
    launches per iteration besides the multiply); the dot products never leave the devices: a kernel
    leaves one partial sum per workgroup, an all_reduce of that slot (4 KiB per dot product) makes every
    rank's partials the element-wise global ones, and the kernel that needs the scalar adds them up.
    Two all_reduce calls per iteration ([p.q], then [r.z, r.r]); the host reads the residual norm every
    `check_every` iterations only.  Same recurrences as ehyb_pcg."""

    def __init__(self, halo_spmv, inv_diag_local=None):
        import torch
        import torch.distributed as dist

        self.torch, self.dist = torch, dist
        self.sh = halo_spmv
        self.L = halo_spmv.L
        self.dev = halo_spmv.x.device
        self.lib = H._lib.load()
        slots, sd, bb, pq, rr, rz0 = (C.c_int() for _ in range(6))
        H._check(self.lib.ehyb_cg_layout(*(C.byref(v) for v in (slots, sd, bb, pq, rr, rz0))), "ehyb_cg_layout")
        self.slot_doubles, self.grid = sd.value, sd.value // 2
        self.A_BB, self.A_PQ, self.A_RR, self.A_RZ0 = bb.value, pq.value, rr.value, rz0.value
        self.s = torch.zeros(slots.value * sd.value, dtype=torch.float64, device=self.dev)
        self.dinv = None
        if inv_diag_local is not None:
            self.dinv = torch.from_numpy(self.L.x_to_plan(np.asarray(inv_diag_local, dtype=np.float64))).to(self.dev)
        # gloo (functional mode on one GPU) cannot reduce device tensors: staged through the host there
        self.stage = self.L.world > 1 and dist.get_backend(self.L.group) != "nccl"

    def _slot(self, first, count=1):
        """`count` neighbouring slots of the partial array (the part of the last one that is written)."""
        sd = self.slot_doubles
        return self.s[first * sd:(first + count - 1) * sd + self.grid]

    def _allreduce(self, view):
        if self.L.world == 1:
            return
        if self.stage:
            h = view.cpu()
            self.dist.all_reduce(h, group=self.L.group)
            view.copy_(h)
        else:
            self.dist.all_reduce(view, group=self.L.group)

    def _scalar(self, slot):
        return float(self.s[slot * self.slot_doubles:slot * self.slot_doubles + self.grid].sum().item())

    def solve(self, b_local, max_iter=1000, rtol=1e-10, check_every=10, x0_local=None):
        """b_local, x0_local: this rank's segments in global label order (host arrays).
        -> (x_local, iterations, relative residual)"""
        torch, L, sh, lib = self.torch, self.L, self.sh, self.lib
        n = L.n_loc
        b = torch.from_numpy(L.x_to_plan(np.asarray(b_local, dtype=np.float64))).to(self.dev)
        x = torch.zeros(n, dtype=torch.float64, device=self.dev)
        if x0_local is not None:
            x.copy_(torch.from_numpy(L.x_to_plan(np.asarray(x0_local, dtype=np.float64))))
        r = torch.empty(n, dtype=torch.float64, device=self.dev)
        p = sh.x[:n]                     # the direction vector lives where the multiply reads it
        q = sh.y
        dinv = self.dinv.data_ptr() if self.dinv is not None else None
        st = lambda: torch.cuda.current_stream().cuda_stream  # noqa: E731
        s_ptr = self.s.data_ptr()
        p.copy_(x)
        sh.step()                                            # q = A x0
        H._check(lib.ehyb_cg_init_step(n, b.data_ptr(), q.data_ptr(), dinv, r.data_ptr(), p.data_ptr(), s_ptr, st()), "ehyb_cg_init_step")
        self._allreduce(self._slot(self.A_BB))
        self._allreduce(self._slot(self.A_RZ0, 2))           # r.z number 0 and r.r are neighbours
        bb = self._scalar(self.A_BB) or 1.0
        rr = self._scalar(self.A_RR)
        it, cur = 0, 0
        while it < max_iter and (rr / bb) ** 0.5 > rtol:
            for _ in range(min(check_every, max_iter - it)):
                sh.step()                                    # q = A p: halo exchange of p + two-phase multiply
                H._check(lib.ehyb_cg_dot_step(n, p.data_ptr(), q.data_ptr(), s_ptr, st()), "ehyb_cg_dot_step")
                self._allreduce(self._slot(self.A_PQ))
                H._check(lib.ehyb_cg_update_step(n, p.data_ptr(), q.data_ptr(), dinv, x.data_ptr(), r.data_ptr(), s_ptr, cur, st()),
                         "ehyb_cg_update_step")
                # the new r.z (number cur ^ 1, slot rz0 + 2 (cur ^ 1)) and r.r (slot rz0 + 1): neighbours either way
                self._allreduce(self._slot(self.A_RZ0 + (cur ^ 1), 2))
                H._check(lib.ehyb_cg_direction_step(n, r.data_ptr(), dinv, p.data_ptr(), s_ptr, cur, st()), "ehyb_cg_direction_step")
                cur ^= 1
                it += 1
            rr = self._scalar(self.A_RR)                     # the only host read in the loop
            if rr != rr:
                raise RuntimeError("HaloCG: breakdown (is the matrix symmetric positive definite?)")
        return L.y_from_plan(x.cpu().numpy()), it, (rr / bb) ** 0.5
