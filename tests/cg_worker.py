"""Worker for tests/test_gpu_dist.py::test_halo_cg_two_ranks: HaloCG (ehyb_spmv_gpu_amd.dist) on one
fem3d block per rank, made positive definite, all ranks on cuda:0 over gloo (functional mode).
Every rank checks its rows of the solution against the true residual of the global system, which
it can evaluate for its own rows from the x segments gathered over gloo."""
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


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    n = 30000
    for sym, jacobi in ((0, False), (1, True)):
        cfg = E.make_config(lds_doubles=2048, sym_pairs=sym)
        m = E.Matrix.generate("fem3d_block", n, 3, 22, 22, 13500, 1, 11, rank, world, cfg=cfg)
        I, J, V = m.I.copy(), m.J.copy(), m.V.copy()
        m.free()
        cuts = [n * r for r in range(world + 1)]
        r0, r1 = cuts[rank], cuts[rank + 1]
        # symmetric + strictly diagonally dominant => positive definite; every rank sees whole rows
        off = np.bincount(I - r0, weights=np.abs(V) * (I != J), minlength=n)
        scale = 1.0 + 50.0 * ((np.arange(r0, r1) * 7919) % 13) if jacobi else np.ones(n)   # badly scaled diagonal for the PCG arm
        V[I == J] = ((off + 1.0) * scale)[(I - r0)[I == J]]
        diag = (off + 1.0) * scale
        L = D.RankLocalMatrix(I, J, V, cuts, rank, cfg, symmetric=True)
        sh = D.HaloSpmv(L, dev, overlap=True, stage_on_cpu=True)
        n_glob = cuts[-1]
        b_glob = O.x_glibc(n_glob) + 0.2
        cg = D.HaloCG(sh, inv_diag_local=1.0 / diag if jacobi else None)
        x_loc, iters, rel = cg.solve(b_glob[r0:r1], max_iter=500, rtol=1e-10, check_every=5)
        # true residual on this rank's rows, from the gathered solution
        parts = [None] * world
        dist.all_gather_object(parts, x_loc)
        x_glob = np.concatenate(parts)
        res = O.spmv_coo(n_glob, I, J, V, x_glob)[r0:r1] - b_glob[r0:r1]
        tot = torch.tensor([float(res @ res), float(b_glob[r0:r1] @ b_glob[r0:r1])], dtype=torch.float64)
        dist.all_reduce(tot)
        true_rel = float((tot[0] / tot[1]) ** 0.5)
        if rank == 0:
            print(f"CG_CASE sym={sym} jacobi={jacobi} world={world} iters={iters} rel={rel:.2e} true_rel={true_rel:.2e}", flush=True)
        assert rel <= 1e-10 and true_rel <= 5e-10 and 0 < iters < 500, (iters, rel, true_rel)
    dist.barrier()
    if rank == 0:
        print("CG_OK", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
