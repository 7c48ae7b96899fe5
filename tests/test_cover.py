"""The "cover" exchange of the multi-GPU step end to end on the CPU, the ranks as threads of one process (dist.ThreadRanks): every rank's
plan walked by the oracle, x entries and partial sums exchanged by hand -- the product of the whole matrix."""
import threading

import numpy as np
import pytest


@pytest.mark.parametrize("world,chunks,cost_model", [(2, 1, 0), (3, 2, 1), (5, 3, 1)])
def test_cover_step_with_the_ranks_as_threads(E, O, world, chunks, cost_model):
    from ehyb_spmv_gpu_amd import dist as D

    cfg = E.make_config(partitioner=E.EHYB_PART_DEGREE, er_panel_cols=512, er_block_rows=300, lds_doubles=256, host_threads=2)
    full = E.Matrix.generate("rmat", 13, 1 << 16, 2, cfg=cfg)
    n = full.n
    x = O.x_glibc(n)
    y_ref = O.spmv_coo(n, full.I, full.J, full.V, x)
    scale = O.abs_rowsum(n, full.I, full.J, full.V, x)
    tr = D.ThreadRanks(world)
    Ls, errs = [None] * world, []

    def build(r):
        try:
            m = E.Matrix.generate("rmat_block", 13, 1 << 16, 2, r, world, cost_model, cfg=cfg)
            cuts = m.block_cuts
            rp = m.row_idx.astype(np.int64)
            a, b = int(rp[cuts[r]]), int(rp[cuts[r + 1]])
            Ls[r] = D.RankLocalMatrix(m.I[a:b].copy(), m.J[a:b].copy(), m.V[a:b].copy(), cuts, r, cfg, group=tr.group(r), exchange="cover", chunks=chunks)
        except Exception as e:  # noqa: BLE001
            errs.append(repr(e))
            tr.barrier.abort()

    ts = [threading.Thread(target=build, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs
    # what travels: never more than all the blocks' columns; the entries handed over arrive somewhere
    assert sum(L.nnz_exported for L in Ls) == sum(L.nnz_imported for L in Ls) > 0 and sum(L.nnz for L in Ls) == full.nnz
    assert all(int(Ls[s].ysend_counts[r]) == int(Ls[r].yrecv_counts[s]) for r in range(world) for s in range(world))
    assert all(int(Ls[s].send_counts[k][r]) == int(Ls[r].recv_counts[k][s]) for r in range(world) for s in range(world) for k in range(chunks))
    # one step: every rank multiplies [own x | the ghost columns its peers sent], ships its foreign rows, adds what it receives
    ys = []
    for L in Ls:
        xe = np.zeros(L.n_loc + L.n_ext)
        xe[:L.n_loc] = L.x_to_plan(x[L.r0:L.r1])
        at = 0
        for k in range(chunks):                                   # chunk k: peer 0's entries, peer 1's, ... = the peers' send lists in order
            col = int(L.col_segs[1 + k])
            for s in range(world):
                S = Ls[s]
                first = int(S.send_first[k]) + int(S.send_counts[k][:L.rank].sum())
                cnt = int(S.send_counts[k][L.rank])
                xe[col:col + cnt] = S.x_to_plan(x[S.r0:S.r1])[S.send_idx[first:first + cnt]]
                col += cnt
                at += cnt
        assert at == L.n_ghost and np.array_equal(xe[L.ghost_slot_col], x[L.ghost_cols])
        y, w = O.walk_plan(L.plan(upload=False), xe)
        ys.append(np.asarray(y))
    for L in Ls:
        yo = ys[L.rank][:L.n_loc].copy()
        at = 0
        for s in range(world):
            S = Ls[s]
            off, cnt = S.n_loc + int(S.ysend_counts[:L.rank].sum()), int(S.ysend_counts[L.rank])
            np.add.at(yo, L.yrecv_idx[at:at + cnt], ys[s][off:off + cnt])
            at += cnt
        bad, worst = O.check_tolerance(L.y_from_plan(yo), y_ref[L.r0:L.r1], scale[L.r0:L.r1])
        assert bad == 0, (L.rank, worst)
