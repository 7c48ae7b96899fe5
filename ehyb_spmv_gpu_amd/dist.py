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


def _setup_device(group=None):
    """Where the tensors of the SETUP collectives live: RCCL moves device tensors, gloo host tensors."""
    import torch
    import torch.distributed as dist

    return torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")


class ThreadRanks:
    """The ranks of a set-up as THREADS of one process (tools/rank_plans_one_gpu.py: every rank's plan of an N-rank decomposition built
    and timed on ONE GPU, one after the other -- the box allows no more than six processes on its card, and no collective is needed to
    time a rank's local multiply).  `ThreadRanks(world).group(rank)` stands in for a torch.distributed group in RankLocalMatrix: alltoallv
    then goes through a mailbox and two barriers.  numpy releases the GIL in its heavy calls, libehyb.so in all of them."""

    def __init__(self, world):
        import threading

        self.world = int(world)
        self.barrier = threading.Barrier(self.world)
        self.box = [None] * self.world

    class Group:
        def __init__(self, ranks, rank):
            self.ranks, self.rank = ranks, int(rank)

    def group(self, rank):
        return ThreadRanks.Group(self, rank)


def alltoallv(send, send_counts, group=None):
    """Uneven all-to-all of a 1-D numpy array with TENSOR collectives (the counts first, then the payload in one
    all_to_all_single): what the ranks tell each other while the send lists are built.  -> (received, recv_counts)"""
    if len(send_counts) == 1:      # one rank (the loop-back test of the RCCL step): what I send is what I receive
        return np.ascontiguousarray(send).copy(), np.asarray([int(send_counts[0])], dtype=np.int64)
    if isinstance(group, ThreadRanks.Group):
        tr, me = group.ranks, group.rank
        send = np.ascontiguousarray(send)
        first = np.concatenate(([0], np.cumsum([int(c) for c in send_counts])))
        tr.box[me] = (send, first)
        tr.barrier.wait()
        parts = [tr.box[s][0][int(tr.box[s][1][me]):int(tr.box[s][1][me + 1])] for s in range(tr.world)]
        out = np.concatenate(parts) if parts else send[:0]
        counts = np.asarray([len(p) for p in parts], dtype=np.int64)
        tr.barrier.wait()          # everybody has read the mailbox: it may be overwritten
        return out, counts
    import torch
    import torch.distributed as dist

    dev = _setup_device(group)
    world = dist.get_world_size(group)
    sc = torch.tensor([int(c) for c in send_counts], dtype=torch.int64, device=dev)
    rc = torch.empty(world, dtype=torch.int64, device=dev)
    dist.all_to_all_single(rc, sc, group=group)
    rc_l = [int(c) for c in rc.cpu().tolist()]
    send_t = torch.from_numpy(np.ascontiguousarray(send)).to(dev)
    recv_t = torch.empty(sum(rc_l), dtype=send_t.dtype, device=dev)
    dist.all_to_all_single(recv_t, send_t, rc_l, [int(c) for c in send_counts], group=group)
    return recv_t.cpu().numpy(), np.asarray(rc_l, dtype=np.int64)


def _even(v):
    return (int(v) + 1) & ~1


class RankLocalMatrix:
    def __init__(self, I, J, V, cuts, rank, cfg=None, symmetric=False, group=None, exchange="halo", chunks=1, chunk_shares=None, loopback=0.0):
        """I, J, V: this rank's rows in global labels, row-grouped (I ascending).  cuts: first row of
        every rank, world+1 entries.  Collective: every rank of `group` must call it.
        exchange = "halo": one ghost slot per distinct remote column, filled by all_to_all steps of exactly
        those entries (HaloSpmv).  The slots come in `chunks` CHUNKS = exchange steps: inside every owner's
        columns the most referenced ones first, chunk k taking chunk_shares[k] of every owner's slots (default: equal
        shares), so that the multiply over chunk k's columns (ehyb_spmv_part) runs while chunk k+1 is on the wire.
        Column layout of the rank's matrix and of x: [own columns | chunk 0: owner 0, owner 1, ... | chunk 1: ... ],
        every part padded to an even length (the column segments of ehyb_plan_create_host_segs).
        exchange = "allgather": the ghost columns are the places of the remote entries inside the buffer an
        all-gather of the (padded) x segments fills (GatherSpmv):
        x = [own segment, padded to seg_len | segment of rank 0 | ... | segment of rank world-1], every
        segment in its owner's plan order, so nothing has to be unpacked after the collective.
        loopback (test device, halo exchange only): the columns of the last `loopback` share of this rank's OWN rows are treated
        as remote columns -- owned by this very rank, which then sends them to itself -- so that a communicator with ONE rank
        exercises pack -> ncclSend / ncclRecv -> ghost columns -> the multiply in parts (RCCL refuses two ranks on one device,
        and the GPU box has one)."""
        world = len(cuts) - 1
        self.rank, self.world, self.cuts, self.group = rank, world, [int(c) for c in cuts], group
        r0, r1 = self.cuts[rank], self.cuts[rank + 1]
        self.r0, self.r1, self.n_loc = r0, r1, r1 - r0
        I = np.asarray(I)
        J = np.asarray(J)
        V = np.asarray(V, dtype=np.float64)
        if len(I) and (I.min() < r0 or I.max() >= r1 or np.any(np.diff(I) < 0)):
            raise ValueError("RankLocalMatrix: rows must lie in [r0, r1) and be grouped in ascending order")
        if exchange not in ("halo", "allgather", "cover"):
            raise ValueError(f"RankLocalMatrix: unknown exchange {exchange!r}")
        loop = float(loopback) > 0 and exchange in ("halo", "cover")
        cover = exchange == "cover" and (world > 1 or loop)
        if exchange == "allgather" or (world == 1 and not loop):
            chunks, chunk_shares = 1, None
        chunks = max(1, int(chunks))
        shares = np.asarray(chunk_shares if chunk_shares is not None else [1.0 / chunks] * chunks, dtype=np.float64)
        if len(shares) != chunks or np.any(shares <= 0):
            raise ValueError("RankLocalMatrix: chunk_shares must hold one positive share per chunk")
        edges = np.cumsum(shares / shares.sum())[:-1]
        own = (J >= r0) & (J < (r1 - int(self.n_loc * float(loopback)) if loop else r1))
        # diagonal block: a masked row-grouped sequence is still row-grouped
        indptr = np.zeros(self.n_loc + 1, dtype=np.int64)
        np.cumsum(np.bincount(I[own] - r0, minlength=self.n_loc), out=indptr[1:])
        cfg1 = _copy_cfg(cfg, n_top=1)
        self.m = H.Matrix.from_csr(indptr, J[own] - r0, V[own], cfg1, symmetric=symmetric)
        self.m.reorder(cfg1)
        self.perm = self.m.reorder_list[:self.n_loc].copy()  # local row i (unpermuted) -> its place in the plan
        # ---- ghost slots: the distinct remote columns, ordered by (chunk, owner, references descending, label)
        off = ~own
        Jg, Ioff, Voff = J[off], I[off], V[off]
        gcols, inv, refs = np.unique(Jg, return_inverse=True, return_counts=True)
        owner = np.searchsorted(np.asarray(self.cuts), gcols, side="right") - 1
        self.n_foreign, self.nnz_exported, self.nnz_imported = 0, 0, 0
        self.ysend_counts = np.zeros(world, dtype=np.int64)
        self.yrecv_counts = np.zeros(world, dtype=np.int64)
        self.yrecv_idx = np.zeros(0, dtype=np.int32)
        foreign = None
        if cover:
            # ---- exchange "cover": per owner p a vertex cover of the block A[my rows, p's columns] -- its columns of degree >= t_p
            # stay GHOST columns (their x entries travel to me), every other entry of the block is handed to p, who multiplies it with
            # its own x and ships one partial sum per row: t_p = the threshold with the fewest columns + rows (t = 1: all columns, the
            # plain halo exchange; t = inf: all rows)
            n_loc1 = max(1, self.n_loc)
            own_e, deg_e = owner[inv], refs[inv]
            key = own_e.astype(np.int64) * n_loc1 + (Ioff - r0)
            ordk = np.argsort(key, kind="stable")
            ks = key[ordk]
            starts = np.flatnonzero(np.concatenate(([True], ks[1:] != ks[:-1]))) if len(ks) else np.zeros(0, dtype=np.int64)
            gmin = np.minimum.reduceat(deg_e[ordk], starts) if len(ks) else np.zeros(0, dtype=np.int64)
            gown = ks[starts] // n_loc1
            best_size, best_t = None, None
            for t in (1, 2, 3, 4, 6, 8, 12, 16, 32, 64, 1 << 30):
                size = np.bincount(owner[refs >= t], minlength=world) + np.bincount(gown[gmin < t], minlength=world)
                if best_size is None:
                    best_size, best_t = size.copy(), np.full(world, t, dtype=np.int64)
                else:
                    better = size < best_size
                    best_size[better], best_t[better] = size[better], t
            self.cover_threshold = best_t
            hot_col = refs >= best_t[owner]
            keep_e = hot_col[inv]
            # the entries that leave, grouped by their new owner: (global row, global column, value)
            out = np.flatnonzero(~keep_e)
            out = out[np.argsort(own_e[out], kind="stable")]
            out_counts = np.bincount(own_e[out], minlength=world).astype(np.int64)
            Ir, cnt_from = alltoallv(Ioff[out].astype(np.int64), out_counts, group)
            Jr, _ = alltoallv(Jg[out].astype(np.int64), out_counts, group)
            Vr, _ = alltoallv(Voff[out].astype(np.float64), out_counts, group)
            self.nnz_exported, self.nnz_imported = int(len(out)), int(len(Ir))
            if len(Jr) and (Jr.min() < r0 or Jr.max() >= r1):
                raise ValueError("RankLocalMatrix: a peer handed over an entry of a column this rank does not own")
            # my FOREIGN rows: one per distinct (origin rank, its row), grouped by origin -- what I ship back every step
            n_glob = int(self.cuts[-1])
            origin = np.repeat(np.arange(world, dtype=np.int64), cnt_from)
            fu, finv = np.unique(origin * n_glob + Ir, return_inverse=True)
            self.n_foreign = len(fu)
            self.ysend_counts = np.bincount(fu // n_glob, minlength=world).astype(np.int64)
            rows_back, self.yrecv_counts = alltoallv(fu % n_glob, self.ysend_counts, group)   # every origin learns which of its rows get a partial sum from me
            if len(rows_back) and (rows_back.min() < r0 or rows_back.max() >= r1):
                raise ValueError("RankLocalMatrix: a peer announced partial sums for rows this rank does not own")
            self.yrecv_idx = self.perm[rows_back - r0].astype(np.int32)
            fo = np.argsort(finv, kind="stable")
            foreign = (finv[fo].astype(np.int32), self.perm[Jr[fo] - r0].astype(np.int32), Vr[fo])
            # what stays here: the entries in the hot columns
            remap = np.cumsum(hot_col) - 1
            Jg, Ioff, Voff, inv = Jg[keep_e], Ioff[keep_e], Voff[keep_e], remap[inv[keep_e]]
            gcols, refs, owner = gcols[hot_col], refs[hot_col], owner[hot_col]
        self.n_ghost = len(gcols)
        by_owner = np.lexsort((gcols, -refs, owner))                       # owner, then hot columns first
        per_owner = np.bincount(owner, minlength=world).astype(np.int64)
        assert loop or per_owner[rank] == 0
        self.exchanges = world > 1 or loop       # there is something to send and receive
        first_of = np.concatenate(([0], np.cumsum(per_owner)))[:-1]
        pos = np.arange(self.n_ghost) - np.repeat(first_of, per_owner)       # rank inside the owner's list
        frac = (pos + 0.5) / np.maximum(1, np.repeat(per_owner, per_owner))
        chunk_sorted = np.searchsorted(edges, frac, side="right")            # chunk of every slot, in by_owner order
        by_chunk = np.argsort(chunk_sorted, kind="stable")                    # chunk major, the rest as it was
        order = by_owner[by_chunk]
        chunk_of = chunk_sorted[by_chunk]
        self.ghost_cols = gcols[order]                                       # global label of every slot, slot order
        owner_s = owner[order]
        self.chunks = chunks
        # recv_counts[k][p]: slots of chunk k whose owner is p
        self.recv_counts = np.zeros((chunks, world), dtype=np.int64)
        np.add.at(self.recv_counts, (chunk_of, owner_s), 1)
        chunk_len = self.recv_counts.sum(axis=1)
        # ---- column segments: own columns, then one per chunk, every start even
        seg = [0, _even(self.n_loc)]
        for k in range(chunks):
            seg.append(_even(seg[-1] + int(chunk_len[k])))
        self.exchange = exchange
        self.seg_len = max(self.cuts[b + 1] - self.cuts[b] for b in range(world))
        chunk_first = np.concatenate(([0], np.cumsum(chunk_len)))[:-1]
        col_of_slot = np.asarray(seg[1:-1], dtype=np.int64)[chunk_of] + (np.arange(self.n_ghost) - chunk_first[chunk_of])
        self.ghost_slot_col = col_of_slot                                    # column (in x) of every slot
        # ---- what the peers want from me, chunk by chunk, in their slot order: tensor collectives
        self.send_counts = np.zeros((chunks, world), dtype=np.int64)
        send_idx = []
        if self.exchanges:
            for k in range(chunks):
                sel = chunk_of == k
                asked, cnt = alltoallv(self.ghost_cols[sel].astype(np.int64), self.recv_counts[k], group)
                if len(asked) and (asked.min() < r0 or asked.max() >= r1):
                    raise ValueError("RankLocalMatrix: a peer asked for a column this rank does not own")
                self.send_counts[k] = cnt
                send_idx.append(self.perm[asked - r0].astype(np.int32))       # where, in my plan order, what each peer wants sits
        self.send_idx = np.concatenate(send_idx) if send_idx else np.zeros(0, np.int32)
        self.send_first = np.concatenate(([0], np.cumsum(self.send_counts.sum(axis=1))))
        slot_of_entry = np.empty(self.n_ghost, dtype=np.int64)
        slot_of_entry[order] = np.arange(self.n_ghost)                       # unique index -> slot
        if exchange == "allgather" and world > 1:
            # every owner tells me where, in ITS plan order, the columns I asked for sit: the reverse all-to-all
            places, _ = alltoallv(self.send_idx.astype(np.int64), self.send_counts[0], group)
            # ghost column of remote entry j of owner p: behind my padded segment, inside p's segment of the gathered buffer
            slot_col = self.n_loc + (self.seg_len - self.n_loc) + owner_s * self.seg_len + places
            self.n_ext = (self.seg_len - self.n_loc) + world * self.seg_len
            self.col_segs = None
            self.m.append_ghosts(self.n_ext, self.perm[I[off] - r0], (slot_col - self.n_loc)[slot_of_entry[inv]], V[off])
        else:
            # the coupling entries, rows in plan numbering, columns = ghost columns behind the own ones
            if cover:   # the matrix is square: room for the foreign rows behind the own ones, the last column segment padded to match
                seg[-1] = max(seg[-1], _even(self.n_loc + self.n_foreign))
            self.n_ext = seg[-1] - self.n_loc
            self.col_segs = np.asarray(seg, dtype=np.int32)
            self.m.append_ghosts(self.n_ext, self.perm[Ioff - r0], (col_of_slot - self.n_loc)[slot_of_entry[inv]], Voff)
            if cover and self.n_foreign:
                cfgr = H.make_config() if cfg is None else cfg
                lib = H._lib.load()
                fi, fj, fv = (np.ascontiguousarray(a) for a in foreign)
                H._check(lib.ehyb_matrix_append_rows(C.byref(self.m.c), self.n_loc, self.n_foreign, len(fi), fi.ctypes.data_as(C.POINTER(C.c_int)),
                                                     fj.ctypes.data_as(C.POINTER(C.c_int)), fv.ctypes.data_as(C.POINTER(C.c_double)), int(cfgr.part_rows)),
                         "ehyb_matrix_append_rows")
        self.nnz = len(V) - self.nnz_exported + self.nnz_imported      # entries this rank multiplies
        self.nnz_own_cols = int(own.sum()) + self.nnz_imported
        self.cover = cover
        # (cover: every window given up -- the foreign rows close in a pass 2 of their own, which only the panel form has)
        kw = dict(row_split=self.n_loc, er_mode=2, prune_pct=1, fuse_er=2, direct=2) if cover else {}
        self.cfg_plan = _copy_cfg(cfg, n_top=2 if self.exchanges else 1, **kw)

    def plan(self, upload=True):
        return H.Plan(self.m, self.cfg_plan, rows=(0, self.n_loc + self.n_foreign), upload=upload,
                      col_segs=self.col_segs if (self.col_segs is not None and self.exchanges) else None)

    def x_to_plan(self, x_local):
        """Local x segment (global label order) -> plan order."""
        return H.vector_reorder(x_local, self.perm)

    def y_from_plan(self, y_plan):
        return H.vector_recover(y_plan, self.perm)


class Comm:
    """RCCL communicator of libehyb.so (include/ehyb.h: ehyb_comm_*): what the C-side step sends and receives through.
    torch.distributed only carries the 128-byte ncclUniqueId from rank 0 to the others (make_comm)."""

    def __init__(self, unique_id, rank, world):
        self.lib = H._lib.load()
        self.rank, self.world = int(rank), int(world)
        h = C.c_void_p()
        H._check(self.lib.ehyb_comm_create(unique_id, self.rank, self.world, C.byref(h)), "ehyb_comm_create")
        self.h = h
        self._halos = []
        st = C.c_void_p()
        H._check(self.lib.ehyb_comm_info(self.h, None, None, C.byref(st)), "ehyb_comm_info")
        self.stream = st.value or 0

    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(128)
        H._check(H._lib.load().ehyb_comm_unique_id(buf), "ehyb_comm_unique_id")
        return buf

    def halo(self, plan, local):
        """ehyb_halo_create for a RankLocalMatrix's plan: the send list and the per-chunk, per-peer counts go to the C side once."""
        idx = np.ascontiguousarray(local.send_idx, dtype=np.int32)
        sc = np.ascontiguousarray(local.send_counts, dtype=np.int64).reshape(-1)
        rc = np.ascontiguousarray(local.recv_counts, dtype=np.int64).reshape(-1)
        h = C.c_void_p()
        H._check(self.lib.ehyb_halo_create(self.h, plan.h, int(local.chunks), idx.ctypes.data_as(C.POINTER(C.c_int32)), len(idx),
                                           sc.ctypes.data_as(C.POINTER(C.c_int64)), rc.ctypes.data_as(C.POINTER(C.c_int64)), C.byref(h)), "ehyb_halo_create")
        self._halos.append(h)
        if getattr(local, "cover", False):
            ys = np.ascontiguousarray(local.ysend_counts, dtype=np.int64)
            yr = np.ascontiguousarray(local.yrecv_counts, dtype=np.int64)
            yi = np.ascontiguousarray(local.yrecv_idx, dtype=np.int32)
            H._check(self.lib.ehyb_halo_set_partials(h, int(local.n_loc), ys.ctypes.data_as(C.POINTER(C.c_int64)), yr.ctypes.data_as(C.POINTER(C.c_int64)),
                                                     yi.ctypes.data_as(C.POINTER(C.c_int32)), len(yi)), "ehyb_halo_set_partials")
        return h

    def allreduce_sum(self, ptr, count, stream=0):
        rc = self.lib.ehyb_comm_allreduce_sum(self.h, ptr, int(count), stream)
        if rc:
            H._check(rc, "ehyb_comm_allreduce_sum")

    def destroy(self):
        for h in self._halos:
            self.lib.ehyb_halo_destroy(h)
        self._halos = []
        if self.h:
            self.lib.ehyb_comm_destroy(self.h)
            self.h = None


def make_comm(group=None):
    """One Comm per rank of `group` (torch.distributed; None with no process group = a single rank): rank 0 draws the
    ncclUniqueId, a broadcast of its 128 bytes hands it to the others, every rank joins (ncclCommInitRank)."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return Comm(Comm.unique_id(), 0, 1)
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    dev = _setup_device(group)
    t = torch.zeros(128, dtype=torch.uint8, device=dev)
    if rank == 0:
        t.copy_(torch.frombuffer(bytearray(Comm.unique_id().raw), dtype=torch.uint8))
    dist.broadcast(t, src=dist.get_global_rank(group, 0) if group else 0, group=group)
    return Comm(C.create_string_buffer(bytes(t.cpu().numpy().tobytes()), 128), rank, world)


class HaloExchange:
    """x_ext = [local x (plan order) | ghost columns, chunk by chunk]; pack() gathers what the peers asked for,
    transfer(k) fills the ghost columns of chunk k.
    mode "a2a" (default): ONE all_to_all_single with uneven splits per chunk (RCCL over xGMI: every pair of GPUs has its
    own link, so the direct all-to-all is link-optimal); "p2p": grouped isend/irecv pairs -- an explicit choice
    (bench.py --exchange-mode p2p), never a silent fallback: a collective that throws is a fault, not a slow path."""

    def __init__(self, local, x_ext, group=None, stage_on_cpu=False, mode="a2a"):
        import torch
        import torch.distributed as dist

        self.torch, self.dist = torch, dist
        self.L, self.group, self.x_ext = local, group, x_ext
        dev = x_ext.device
        self.send_idx = torch.from_numpy(local.send_idx.astype(np.int32)).to(dev)
        self.send_buf = torch.empty(len(local.send_idx), dtype=torch.float64, device=dev)
        self.send_counts = [[int(c) for c in row] for row in local.send_counts]
        self.recv_counts = [[int(c) for c in row] for row in local.recv_counts]
        if mode not in ("a2a", "p2p"):
            raise ValueError(f"HaloExchange: unknown mode {mode!r}")
        self.mode = mode
        self.stage = stage_on_cpu and dev.type != "cpu"  # gloo cannot move GPU tensors: through the host (functional mode)
        segs = local.col_segs
        self.ghost_views = [x_ext[int(segs[k + 1]):int(segs[k + 1]) + sum(self.recv_counts[k])] for k in range(local.chunks)] if local.exchanges else []
        self.send_views = [self.send_buf[int(local.send_first[k]):int(local.send_first[k + 1])] for k in range(local.chunks)] if local.exchanges else []
        self.lib = H._lib.load()
        self._pack_args = (C.c_void_p(self.x_ext.data_ptr()), C.c_void_p(self.send_idx.data_ptr()), C.c_void_p(self.send_buf.data_ptr()), len(local.send_idx))

    def pack(self, compute_stream=0, comm_stream=0):
        """send_buf[i] = x[send_idx[i]] on compute_stream (one launch for all chunks); comm_stream then waits for it."""
        rc = self.lib.ehyb_step_pack(self._pack_args[0], self._pack_args[1], self._pack_args[2], self._pack_args[3], compute_stream, comm_stream)
        if rc:
            H._check(rc, "ehyb_step_pack")

    def transfer(self, k):
        L, dist, torch = self.L, self.dist, self.torch
        if not L.exchanges:
            return
        send, recv = self.send_views[k], self.ghost_views[k]
        if L.world == 1:            # loop-back: the rank is its own peer -- a device copy stands in for the collective
            recv.copy_(send)
            return
        sc, rc = self.send_counts[k], self.recv_counts[k]
        if self.stage:
            send, recv_dev, recv = send.cpu(), recv, torch.empty(sum(rc), dtype=torch.float64)
        if self.mode == "a2a":
            dist.all_to_all_single(recv, send, rc, sc, group=self.group)
        else:
            ops, so, ro = [], 0, 0
            for q in range(L.world):
                peer = dist.get_global_rank(self.group, q) if self.group else q
                if sc[q]:
                    ops.append(dist.P2POp(dist.isend, send[so:so + sc[q]], peer, group=self.group))
                if rc[q]:
                    ops.append(dist.P2POp(dist.irecv, recv[ro:ro + rc[q]], peer, group=self.group))
                so += sc[q]
                ro += rc[q]
            for r in (dist.batch_isend_irecv(ops) if ops else []):
                r.wait()
        if self.stage:
            recv_dev.copy_(recv)

    def setup_partials(self, y):
        """exchange "cover": the buffers of the partial-sum exchange -- y[n_loc:] (this rank's foreign rows, grouped by destination)
        against ybuf (what the others computed for this rank's rows) + the rows they belong to."""
        L, torch = self.L, self.torch
        self.y = y
        self.ybuf = torch.zeros(max(1, int(L.yrecv_counts.sum())), dtype=torch.float64, device=y.device)
        self.yidx = torch.from_numpy(L.yrecv_idx.astype(np.int32)).to(y.device)
        self.ysend_counts = [int(c) for c in L.ysend_counts]
        self.yrecv_counts = [int(c) for c in L.yrecv_counts]

    def transfer_partials(self):
        L, dist, torch = self.L, self.dist, self.torch
        n_send, n_recv = sum(self.ysend_counts), sum(self.yrecv_counts)
        send, recv = self.y[L.n_loc:L.n_loc + n_send], self.ybuf[:n_recv]
        if L.world == 1:            # loop-back
            recv.copy_(send)
            return
        if self.stage:
            send, recv_dev, recv = send.cpu(), recv, torch.empty(n_recv, dtype=torch.float64)
        dist.all_to_all_single(recv, send.contiguous(), self.yrecv_counts, self.ysend_counts, group=self.group)
        if self.stage:
            recv_dev.copy_(recv)

    def add_partials(self, stream=0):
        """y[row] += partial for everything transfer_partials delivered (ehyb_scatter_add)."""
        n = sum(self.yrecv_counts)
        rc = self.lib.ehyb_scatter_add(C.c_void_p(self.y.data_ptr()), C.c_void_p(self.yidx.data_ptr()), C.c_void_p(self.ybuf.data_ptr()), n, stream)
        if rc:
            H._check(rc, "ehyb_scatter_add")

    def run(self):
        cur = self.torch.cuda.current_stream().cuda_stream if self.x_ext.device.type == "cuda" else 0
        self.pack(cur, cur)
        for k in range(self.L.chunks):
            self.transfer(k)


class HaloSpmv:
    """One rank's multiply of a RankLocalMatrix on its GPU: y_loc = A[r0:r1, :] x.
    pipelined (default): the step is  pack | own columns (ELL + the panels of the rank's own columns)  ||  chunk 0 on the
    wire | chunk 0's panels  ||  chunk 1 on the wire | ... | closing pass -- every part one C call (ehyb_step_part), the
    collectives on a side stream.  pipelined=False: pack, every chunk, then the whole multiply in one call (the plain step
    the pipelined one is checked against)."""

    def __init__(self, local, device, overlap=True, stage_on_cpu=False, mode="a2a", c_step=False, comm=None):
        """comm: a Comm (RCCL communicator of libehyb.so, make_comm) -- the step is then ONE C call, ehyb_halo_spmv: pack, the
        chunks as grouped ncclSend / ncclRecv pairs on the communicator's stream, the parts of the multiply behind them; torch
        takes no part in it.  Without one the collectives are torch.distributed's, issued from Python (the A/B arm, and the
        only way over gloo).
        c_step: the pipelined step through ONE C call (ehyb_halo_step) that calls back for every collective -- here the
        callback is Python, so it saves nothing and serves as the test of that entry point."""
        import torch

        self.torch = torch
        self.L = local
        self.c_step = c_step
        self.plan = local.plan()
        self.x = torch.zeros(local.n_loc + local.n_ext, dtype=torch.float64, device=device)
        self.cover = bool(getattr(local, "cover", False))
        self.y = torch.zeros(local.n_loc + (local.n_foreign if self.cover else 0), dtype=torch.float64, device=device)   # own rows | foreign rows
        self.halo = HaloExchange(local, self.x, local.group, stage_on_cpu=stage_on_cpu, mode=mode)
        if self.cover:
            if self.plan.stats["nnz_ell"] != 0 or self.plan.stats["er_partials"] == 0:
                raise RuntimeError("HaloSpmv: the cover exchange needs a plan in panel form alone")
            self.halo.setup_partials(self.y)
        self.comm, self.c_halo = comm, None
        if comm is not None and local.exchanges:
            if comm.world != local.world or comm.rank != local.rank:
                raise ValueError("HaloSpmv: the communicator and the matrix disagree about rank / world size")
            self.c_halo = comm.halo(self.plan, local)
        self.overlap = overlap and local.exchanges
        self.comm_stream = torch.cuda.Stream(device=device) if self.overlap else None
        st = self.plan.stats
        # a plan that multiplies in one launch (inline residual / direct shape) has no parts: plain steps only
        self.has_parts = not (st["er_inline"] > 0)
        self.lib = H._lib.load()
        self._xp, self._yp = C.c_void_p(self.x.data_ptr()), C.c_void_p(self.y.data_ptr())

    def set_x_local(self, x_local):
        """x_local: this rank's x segment in global label order."""
        self.x[:self.L.n_loc].copy_(self.torch.from_numpy(self.L.x_to_plan(x_local)))

    def _part(self, cur, comm, wait, s0, s1, flags):
        # (x and y never move: their pointers are wrapped once -- the host side of a step is as dear as the device side)
        rc = self.lib.ehyb_step_part(self.plan.h, self._xp, self._yp, cur, comm, wait, s0, s1, flags)
        if rc:
            H._check(rc, "ehyb_step_part")

    def step(self):
        torch = self.torch
        cur_s = torch.cuda.current_stream()
        cur = cur_s.cuda_stream
        xp, yp = self.x.data_ptr(), self.y.data_ptr()
        L = self.L
        if self.c_halo is not None:
            rc = self.lib.ehyb_halo_spmv(self.c_halo, self._xp, self._yp, cur)
            if rc:
                H._check(rc, "ehyb_halo_spmv")
            return
        if not L.exchanges:
            self.plan.spmv(xp, yp, cur)
            return
        if not (self.overlap and self.has_parts):
            self.halo.pack(cur, cur)
            for k in range(L.chunks):
                self.halo.transfer(k)
            self.plan.spmv(xp, yp, cur)
            if self.cover:                     # the plain cover step: everything multiplied, then the partial sums travel and are added
                self.halo.transfer_partials()
                self.halo.add_partials(cur)
            return
        comm = self.comm_stream.cuda_stream
        K = L.chunks
        if self.cover:
            # own columns (own AND foreign rows) + the close of the foreign rows; their partial sums travel while the chunks multiply
            self.halo.pack(cur, comm)
            with torch.cuda.stream(self.comm_stream):
                for k in range(K):
                    self.halo.transfer(k)
            self._part(cur, comm, 0, 0, 1, 1 | 4)                    # EHYB_PART_FIRST | EHYB_PART_LAST_FOREIGN
            self.comm_stream.wait_stream(cur_s)
            with torch.cuda.stream(self.comm_stream):
                self.halo.transfer_partials()
            for k in range(K):
                self._part(cur, comm, 1 if k == 0 else 0, 1 + k, 2 + k, 2 if k == K - 1 else 0)   # (one wait for everything the side stream holds)
            self.halo.add_partials(cur)
            return
        if self.c_step:
            if not hasattr(self, "_cb"):
                def exchange(k, _comm, _user):
                    try:
                        self.halo.transfer(k)
                        return 0
                    except Exception:          # noqa: BLE001  (reported by the C side as EHYB_ERR_STATE)
                        return 1
                self._cb = C.CFUNCTYPE(C.c_int, C.c_int, C.c_void_p, C.c_void_p)(exchange)
            a = self.halo._pack_args
            with torch.cuda.stream(self.comm_stream):
                H._check(self.lib.ehyb_halo_step(self.plan.h, self._xp, self._yp, a[1], a[2], a[3], K, C.cast(self._cb, C.c_void_p), None, cur, comm),
                         "ehyb_halo_step")
            return
        self.halo.pack(cur, comm)                      # comm waits for the packed send buffer
        self._part(cur, comm, 0, 0, 1, 1)              # own columns: ELL + their panels, while chunk 0 travels
        with torch.cuda.stream(self.comm_stream):
            for k in range(K):
                self.halo.transfer(k)                  # enqueued on comm (behind chunk k-1)
                # compute waits for what comm holds NOW (chunks <= k), then multiplies chunk k's columns
                self._part(cur, comm, 1, 1 + k, 2 + k, 2 if k == K - 1 else 0)

    def y_local(self):
        """This rank's y segment in global label order (host array)."""
        return self.L.y_from_plan(self.y[:self.L.n_loc].cpu().numpy())

    def time_local(self, iters):
        return _time_local(self, iters)

    def time_parts(self, iters):
        """Mean milliseconds of (the part that needs own columns only, the whole local multiply): how much of the local
        work can run while the first chunk is on the wire."""
        torch = self.torch
        cur = torch.cuda.current_stream()
        if not self.has_parts or not self.L.exchanges:
            return 0.0, _time_local(self, iters)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(3):
            self._part(cur.cuda_stream, cur.cuda_stream, 0, 0, 1, 1)
        a.record(cur)
        for _ in range(iters):
            self._part(cur.cuda_stream, cur.cuda_stream, 0, 0, 1, 1)
        b.record(cur)
        b.synchronize()
        return a.elapsed_time(b) / iters, _time_local(self, iters)


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

    def __init__(self, local, device, overlap=True, stage_on_cpu=False, comm=None):
        """comm: a Comm -- the step is then ONE C call (ehyb_gather_spmv: ncclAllGather on the communicator's stream beside the ELL
        phase, then the residual phase)."""
        import torch
        import torch.distributed as dist

        self.comm = comm if (comm is not None and local.world > 1) else None
        self.lib = H._lib.load()
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
        if self.comm is not None:
            rc = self.lib.ehyb_gather_spmv(self.comm.h, self.plan.h, xp, yp, self.seg_len, cur.cuda_stream)
            if rc:
                H._check(rc, "ehyb_gather_spmv")
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
        if getattr(self.sh, "comm", None) is not None:
            # RCCL from C, on the stream the vector kernels run on: in order with them, no host round trip
            self.sh.comm.allreduce_sum(view.data_ptr(), view.numel(), self.torch.cuda.current_stream().cuda_stream)
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
        q = sh.y[:n]                     # (the exchange "cover" keeps the rank's foreign rows behind its own)
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
