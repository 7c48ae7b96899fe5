"""Worker for tests/test_gpu_dist.py::test_pipelined_step_equals_plain_step: every rank on cuda:0 over gloo (functional
mode), rank-local build of one R-MAT (and one FEM matrix) with the ghost columns in several chunks; the PIPELINED step
(own columns | chunk 0 | chunk 1 | ... each multiplied as it lands, collectives on a side stream) against the PLAIN step
(pack, every chunk, then the whole multiply in one call) on the same plan, and both against the CPU oracle."""
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


def case(tag, I, J, V, cuts, rank, world, cfg, chunks, shares, symmetric, dev, mode="a2a", exact=False, c_step=False, exchange="halo"):
    n = cuts[-1]
    r0, r1 = cuts[rank], cuts[rank + 1]
    x = O.x_glibc(n)
    y_ref = O.spmv_coo(n, I, J, V, x)[r0:r1]
    scale = O.abs_rowsum(n, I, J, V, x)[r0:r1]
    L = D.RankLocalMatrix(I, J, V, cuts, rank, cfg, symmetric=symmetric, chunks=chunks, chunk_shares=shares, exchange=exchange)
    sh = D.HaloSpmv(L, dev, overlap=True, stage_on_cpu=True, mode=mode, c_step=c_step)
    sh.set_x_local(x[r0:r1])
    st = sh.plan.stats
    ys = []
    for pipelined in (True, False, True):
        sh.overlap = pipelined
        sh.y.fill_(float("nan"))
        sh.x[L.n_loc:].zero_()                   # the ghost columns must come from this step's exchange
        sh.step()
        torch.cuda.synchronize()
        ys.append(sh.y_local())
    bad = 0
    for y in ys:
        b, worst = O.check_tolerance(y, y_ref, scale)
        bad += b
    same = bool(np.array_equal(ys[0], ys[1]) and np.array_equal(ys[0], ys[2]))
    gap = float(np.max(np.abs(ys[0] - ys[1]) / np.maximum(scale, 1e-300))) if len(scale) else 0.0
    # the two schedules launch the same kernels on the same data: the results can differ only where LDS / global atomics
    # add in a different order (panel form: pass 2; split rows) -- rounding level, far inside the 1e-12 tolerance
    if exact and not same:
        bad += 1
    if gap > 1e-14:
        bad += 1
    t = torch.tensor([float(bad), float(same), float(st["er_partials"]), gap], dtype=torch.float64)
    mx = t.clone()
    dist.all_reduce(t)
    dist.all_reduce(mx, op=dist.ReduceOp.MAX)
    if rank == 0:
        print(f"PIPE_CASE {tag} world={world} chunks={L.chunks} segs={sh.plan.col_segs} partials_total={int(t[2])} bitwise_same_ranks={int(t[1])}/{world} "
              f"max_gap={mx[3].item():.2e} bad={int(t[0])}", flush=True)
    return int(t[0])


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    bad = 0
    # R-MAT 2^18, every rank generates its block: the panel form, every window given up (what bench.py --gpus N runs)
    cfg = E.make_config(partitioner=E.EHYB_PART_AUTO)
    m = E.Matrix.generate("rmat_block", 18, 1 << 21, 1, rank, world, cfg=cfg)
    cuts = m.block_cuts
    rp = m.row_idx.astype(np.int64)
    a, b = int(rp[cuts[rank]]), int(rp[cuts[rank + 1]])
    I, J, V = m.I[a:b].copy(), m.J[a:b].copy(), m.V[a:b].copy()
    m.free()
    cfgp = E.make_config(er_mode=2, er_panel_cols=4096)
    bad += case("rmat-18-panel-3chunks", I, J, V, cuts, rank, world, cfgp, 3, [0.2, 0.3, 0.5], False, dev)
    bad += case("rmat-18-panel-2chunks-p2p", I, J, V, cuts, rank, world, cfgp, 2, None, False, dev, mode="p2p")
    bad += case("rmat-18-csr-2chunks", I, J, V, cuts, rank, world, E.make_config(er_mode=1), 2, None, False, dev)
    # the same pipelined step through the one-call C entry point (ehyb_halo_step) with the collectives as callbacks
    bad += case("rmat-18-panel-3chunks-c-step", I, J, V, cuts, rank, world, cfgp, 3, [0.2, 0.3, 0.5], False, dev, c_step=True)
    # exchange "cover": hub columns travel as x, the rest of every block was handed to the columns' owner and comes back as one partial
    # sum per row -- the foreign rows close early (EHYB_PART_LAST_FOREIGN) and travel while the chunks multiply
    bad += case("rmat-18-cover-2chunks", I, J, V, cuts, rank, world, E.make_config(er_panel_cols=4096), 2, [0.25, 0.75], False, dev, exchange="cover")
    # a FEM matrix cut into slabs: windows kept, small CSR residual over the ghost columns -- deterministic kernels: bit for bit
    n = 30000
    cfgf = E.make_config(lds_doubles=4096)
    mf = E.Matrix.generate("fem3d_block", n, 3, 22, 22, 13500, 1, 7, rank, world, cfg=cfgf)
    fc = [n * r for r in range(world + 1)]
    bad += case("fem3d-windows-2chunks", mf.I.copy(), mf.J.copy(), mf.V.copy(), fc, rank, world, cfgf, 2, None, False, dev, exact=True)
    dist.barrier()
    if rank == 0:
        print("PIPE_OK" if bad == 0 else f"PIPE_FAIL {bad}", flush=True)
    dist.destroy_process_group()
    sys.exit(0 if bad == 0 else 1)


if __name__ == "__main__":
    main()
