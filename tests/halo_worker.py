"""Worker for tests/test_distributed_cpu.py::test_halo_path_gloo: the rank-local build and the
halo exchange (ehyb_spmv_gpu_amd.dist: RankLocalMatrix, HaloExchange) under gloo on CPU.  The
multiply itself is the oracle's CPU walk of the rank's plan over x = [local x | ghost slots]."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import ehyb_spmv_gpu_amd as E  # noqa: E402
from ehyb_spmv_gpu_amd import dist as D  # noqa: E402
from oracle import oracle as O  # noqa: E402


def check_rank(tag, I, J, V, cuts, rank, world, cfg, symmetric, chunks=1, shares=None):
    n_glob = cuts[-1]
    r0, r1 = cuts[rank], cuts[rank + 1]
    x = O.x_glibc(n_glob)                       # every rank can compute any x entry: x is a function of the index
    y_ref = O.spmv_coo(n_glob, I, J, V, x)[r0:r1]
    scale = O.abs_rowsum(n_glob, I, J, V, x)[r0:r1]
    L = D.RankLocalMatrix(I, J, V, cuts, rank, cfg, symmetric=symmetric, chunks=chunks, chunk_shares=shares)
    assert L.m.n == L.n_loc + L.n_ext and L.m.nnz == len(V)
    assert L.recv_counts.shape == (L.chunks, world) and L.send_counts.shape == (L.chunks, world)
    assert int(L.recv_counts.sum()) == L.n_ghost and not L.recv_counts[:, rank].any() and not L.send_counts[:, rank].any()
    segs = L.col_segs
    assert segs[0] == 0 and segs[-1] == L.m.n and len(segs) == L.chunks + 2 and not (segs[1:-1] & 1).any() and segs[1] >= L.n_loc
    # inside a chunk the slots are grouped by owner, and an owner's columns come hottest first
    gs, refs = np.unique(J[(J < r0) | (J >= r1)], return_counts=True)
    assert len(gs) == L.n_ghost and np.array_equal(np.sort(L.ghost_cols), gs)
    slot_refs = refs[np.searchsorted(gs, L.ghost_cols)]
    owner = np.searchsorted(np.asarray(cuts), L.ghost_cols, side="right") - 1
    at = 0
    for k in range(L.chunks):
        for p in range(world):
            c = int(L.recv_counts[k, p])
            assert np.all(owner[at:at + c] == p) and np.all(np.diff(slot_refs[at:at + c]) <= 0)
            at += c
    assert at == L.n_ghost
    plan = L.plan(upload=False)
    if world > 1:
        assert plan.col_segs == L.chunks + 1 and np.array_equal(plan.array("col_seg_first"), segs)
    st = plan.stats
    # phase 1 must not touch a ghost slot: window columns are local
    hc = plan.array("halo_cols")
    assert len(hc) == 0 or hc.max() < L.n_loc
    assert st["er_inline"] == 0 or world == 1
    # every entry that references a remote column is in the residual
    remote = int(((J < r0) | (J >= r1)).sum())
    assert st["nnz_er"] >= remote and (world > 1 or remote == 0)
    x_ext = torch.zeros(L.n_loc + L.n_ext, dtype=torch.float64)
    x_ext[:L.n_loc] = torch.from_numpy(L.x_to_plan(x[r0:r1]))
    if world > 1:
        # pack on the CPU (ehyb_step_pack is a device call): what the peers asked for, in their slot order
        hx = D.HaloExchange(L, x_ext, mode="p2p" if tag.endswith("-p2p") else "a2a")
        hx.send_buf.copy_(x_ext[torch.from_numpy(L.send_idx.astype(np.int64))])
        for k in range(L.chunks):
            hx.transfer(k)
    assert np.array_equal(x_ext.numpy()[L.ghost_slot_col], x[L.ghost_cols]), "ghost slots do not hold the owners' x entries"
    if tag == "rmat-rows-panel" and world > 1:
        u2 = plan.array("pb_units2").reshape(-1, 4)
        assert st["er_partials"] > 0 and np.any(u2[:, 3] < 0), "the panel case should run in assign mode"
    if st["er_partials"] > 0 and world > 1:
        # a panel never straddles a column segment, and the units of a segment are a run of the unit list
        u1 = plan.array("pb_units1").reshape(-1, 4)
        it1 = plan.array("pb_items1").reshape(-1, 2)
        si = plan.array("pb_seg_item")
        assert len(si) == len(segs) and si[0] == 0 and si[-1] == len(it1)
        for s in range(len(segs) - 1):
            if si[s + 1] > si[s]:
                uu = u1[it1[si[s], 0]:it1[si[s + 1] - 1, 1]]
                assert np.all(uu[:, 0] >= segs[s]) and np.all(uu[:, 0] + uu[:, 1] <= segs[s + 1])
    y_plan, written = O.walk_plan(plan, x_ext.numpy())
    assert written[:L.n_loc].min() == 1 and written.sum() == L.n_loc
    y = L.y_from_plan(y_plan[:L.n_loc])
    bad, worst = O.check_tolerance(y, y_ref, scale)
    # the same rows with the all-gather layout: ghost columns = places inside the gathered segments
    G = D.RankLocalMatrix(I, J, V, cuts, rank, cfg, symmetric=symmetric, exchange="allgather")
    seg = G.seg_len
    assert seg == max(cuts[b + 1] - cuts[b] for b in range(world)) and G.n_ghost == L.n_ghost
    if world > 1:
        assert G.m.n == seg * (world + 1)
        xg = torch.zeros(seg * (world + 1), dtype=torch.float64)
        xg[:G.n_loc] = torch.from_numpy(G.x_to_plan(x[r0:r1]))
        dist.all_gather_into_tensor(xg[seg:], xg[:seg].clone())
        gplan = G.plan(upload=False)
        ghc = gplan.array("halo_cols")
        assert len(ghc) == 0 or ghc.max() < G.n_loc
        yg, wg = O.walk_plan(gplan, xg.numpy())
        assert wg[:G.n_loc].min() == 1 and wg.sum() == G.n_loc
        bad_g, _ = O.check_tolerance(G.y_from_plan(yg[:G.n_loc]), y_ref, scale)
        bad += bad_g
    tot = torch.tensor([float(L.n_ghost), float(len(V)), float(bad)], dtype=torch.float64)
    dist.all_reduce(tot)
    if rank == 0:
        print(f"HALO_CASE {tag} world={world} ghosts_total={int(tot[0])} nnz_total={int(tot[1])} bad={int(tot[2])} worst0={worst:.2e}", flush=True)
    return int(tot[2])


def check_cover(tag, I, J, V, cuts, rank, world, cfg, chunks=1, shares=None, loopback=0.0):
    """exchange "cover": per pair of ranks the hub columns of the block travel as x entries, the rest of the block is handed to the
    columns' owner, who ships one partial sum per row.  The rank's plan (own rows + foreign rows, all in panel form) walked by the
    oracle over x = [own | ghosts], the partial sums exchanged over gloo and added: the oracle's product on the rank's rows."""
    n_glob = cuts[-1]
    r0, r1 = cuts[rank], cuts[rank + 1]
    x = O.x_glibc(n_glob)
    y_ref = O.spmv_coo(n_glob, I, J, V, x)[r0:r1]
    scale = O.abs_rowsum(n_glob, I, J, V, x)[r0:r1]
    R = D.RankLocalMatrix(I, J, V, cuts, rank, cfg, chunks=chunks, chunk_shares=shares, loopback=loopback)                      # the plain halo exchange, for the volumes
    L = D.RankLocalMatrix(I, J, V, cuts, rank, cfg, exchange="cover", chunks=chunks, chunk_shares=shares, loopback=loopback)
    assert L.cover and L.n_ghost <= R.n_ghost
    assert L.n_ghost + int(L.yrecv_counts.sum()) <= R.n_ghost, "a cover never moves more than all the block's columns"
    assert int(L.ysend_counts.sum()) == L.n_foreign and len(L.yrecv_idx) == int(L.yrecv_counts.sum())
    assert L.m.n == L.n_loc + L.n_ext and L.col_segs[-1] == L.m.n and L.m.n >= L.n_loc + L.n_foreign
    plan = L.plan(upload=False)
    st = plan.stats
    assert st["nnz_ell"] == 0 and st["n_rows"] == L.n_loc + L.n_foreign and st["nnz"] == L.nnz
    u2 = plan.array("pb_units2").reshape(-1, 4)
    assert not np.any((u2[:, 2] < L.n_loc) & (u2[:, 2] + np.abs(u2[:, 3]) > L.n_loc)), "a row block straddles the own / foreign boundary"
    # the foreign rows have entries in the rank's OWN columns only: they are complete after column segment 0
    rp = L.m.row_idx.astype(np.int64)
    assert L.n_foreign == 0 or L.m.J[rp[L.n_loc]:rp[L.n_loc + L.n_foreign]].max() < L.n_loc
    x_ext = torch.zeros(L.n_loc + L.n_ext, dtype=torch.float64)
    x_ext[:L.n_loc] = torch.from_numpy(L.x_to_plan(x[r0:r1]))
    hx = D.HaloExchange(L, x_ext)
    hx.send_buf.copy_(x_ext[torch.from_numpy(L.send_idx.astype(np.int64))])
    for k in range(L.chunks):
        hx.transfer(k)
    assert np.array_equal(x_ext.numpy()[L.ghost_slot_col], x[L.ghost_cols])
    y_plan, written = O.walk_plan(plan, x_ext.numpy())
    assert written[:L.n_loc + L.n_foreign].min() == 1
    y_t = torch.from_numpy(np.ascontiguousarray(y_plan[:L.n_loc + L.n_foreign]))
    hx.setup_partials(y_t)
    hx.transfer_partials()
    y_own = y_t[:L.n_loc].numpy().copy()
    np.add.at(y_own, L.yrecv_idx, hx.ybuf.numpy()[:len(L.yrecv_idx)])
    bad, worst = O.check_tolerance(L.y_from_plan(y_own), y_ref, scale)
    tot = torch.tensor([float(R.n_ghost), float(L.n_ghost + int(L.yrecv_counts.sum())), float(L.nnz_exported), float(bad)], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tot)
    if rank == 0:
        print(f"HALO_CASE {tag} world={world} doubles_received: all columns {int(tot[0])} -> cover {int(tot[1])}; entries handed over {int(tot[2])} bad={int(tot[3])} worst0={worst:.2e}",
              flush=True)
    return int(tot[3])


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    bad = 0
    # 1. weak scaling: one fem3d block per rank, each rank generates its own rows only
    n = 9000
    cfg = E.make_config(lds_doubles=512)
    m = E.Matrix.generate("fem3d_block", n, 3, 12, 12, 20000, 1, 7, rank, world, cfg=cfg)
    cuts = [n * r for r in range(world + 1)]
    bad += check_rank("fem3d_block", m.I.copy(), m.J.copy(), m.V.copy(), cuts, rank, world, cfg, symmetric=False)
    # 1b. the same with symmetric pair storage inside every rank's diagonal block (ghost columns untouched)
    cfg = E.make_config(lds_doubles=1024, sym_pairs=1)
    m = E.Matrix.generate("fem3d_block", n, 3, 12, 12, 20000, 1, 7, rank, world, cfg=cfg)
    bad += check_rank("fem3d_block-sym", m.I.copy(), m.J.copy(), m.V.copy(), cuts, rank, world, cfg, symmetric=True)
    # 2. a matrix with no locality cut into ragged row ranges: nearly every column is a ghost somewhere
    cfg = E.make_config(lds_doubles=256)
    g = E.Matrix.generate("rmat", 12, 1 << 15, 2, cfg=cfg)
    base = [0, 1500, 2600, 4096][:world] + [g.n] if world <= 3 else [g.n * r // world for r in range(world + 1)]
    rp = g.row_idx
    a, b = int(rp[base[rank]]), int(rp[base[rank + 1]])
    bad += check_rank("rmat-rows", g.I[a:b].copy(), g.J[a:b].copy(), g.V[a:b].copy(), base, rank, world, cfg, symmetric=False)
    # 2b. the same rows with the residual in panel form: what `bench.py --gpus N` runs on R-MAT 2^24 -- no window of such a
    # rank pays, every partition is given up, phase 1 is empty and pass 2 of the panel residual assigns every row
    cfgp = E.make_config(lds_doubles=256, er_mode=2, er_panel_cols=512, er_block_rows=300)
    bad += check_rank("rmat-rows-panel", g.I[a:b].copy(), g.J[a:b].copy(), g.V[a:b].copy(), base, rank, world, cfgp, symmetric=False)
    # 2c. the same with the ghost columns in three chunks of unequal share (three exchange steps, four column segments),
    # and in two chunks through explicit send/recv pairs
    bad += check_rank("rmat-rows-panel-chunks", g.I[a:b].copy(), g.J[a:b].copy(), g.V[a:b].copy(), base, rank, world, cfgp, symmetric=False,
                      chunks=3, shares=[0.2, 0.3, 0.5])
    bad += check_rank("rmat-rows-chunks-p2p", g.I[a:b].copy(), g.J[a:b].copy(), g.V[a:b].copy(), base, rank, world, cfg, symmetric=False, chunks=2)
    # 2d. exchange "cover" on the same rows: hub columns as x, the rest as partial sums (world 1: the rank is its own peer)
    cfgc = E.make_config(lds_doubles=256, er_panel_cols=512, er_block_rows=300)
    lb = 0.5 if world == 1 else 0.0
    bad += check_cover("rmat-rows-cover", g.I[a:b].copy(), g.J[a:b].copy(), g.V[a:b].copy(), base, rank, world, cfgc, loopback=lb)
    bad += check_cover("rmat-rows-cover-chunks", g.I[a:b].copy(), g.J[a:b].copy(), g.V[a:b].copy(), base, rank, world, cfgc, chunks=3, shares=[0.2, 0.3, 0.5], loopback=lb)
    # 3. block diagonal: no ghosts at all, the exchange is empty
    cfg = E.make_config(lds_doubles=1024, window_mode=1, partitioner=1)
    g = E.Matrix.generate("banded", 2048, 16, 1024, cfg=cfg)
    cuts = [2048 * r for r in range(world + 1)]
    bad += check_rank("block-diagonal", g.I + 2048 * rank, g.J + 2048 * rank, g.V.copy(), cuts, rank, world, cfg, symmetric=False)
    # 4. fuzz: random matrices (tests/fuzz_cases.py) cut into ragged row ranges, random plan configurations
    from fuzz_cases import random_config_kwargs, random_matrix

    for seed in range(300, 308):
        rng = np.random.default_rng(seed)            # the same stream on every rank
        A = random_matrix(rng)
        while A.shape[0] < 2 * world:
            A = random_matrix(rng)
        kw = random_config_kwargs(rng)
        nA = A.shape[0]
        inner = np.sort(rng.choice(np.arange(1, nA), size=world - 1, replace=False)) if world > 1 else np.zeros(0, int)
        fcuts = [0] + [int(c) for c in inner] + [nA]
        a, b = int(A.indptr[fcuts[rank]]), int(A.indptr[fcuts[rank + 1]])
        rows = np.repeat(np.arange(nA, dtype=np.int32), np.diff(A.indptr))
        bad += check_rank(f"fuzz-{seed}", rows[a:b].copy(), A.indices[a:b].astype(np.int32), A.data[a:b].copy(), fcuts, rank, world,
                          E.make_config(**kw), symmetric=False, chunks=1 + seed % 3)
    dist.barrier()
    if rank == 0:
        print("HALO_OK" if bad == 0 else f"HALO_FAIL {bad}", flush=True)
    dist.destroy_process_group()
    sys.exit(0 if bad == 0 else 1)


if __name__ == "__main__":
    main()
