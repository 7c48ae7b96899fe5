"""Shared helpers for the parity tests: the harness flow of solver_test.c:350-389."""
import socket
import subprocess
import sys

import numpy as np


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def torchrun(world, script, script_args=(), timeout=600, env=None, cwd=None):
    """`python -m torch.distributed.run` of `script` with `world` ranks on 127.0.0.1.  The rendezvous port is picked by binding port 0
    and letting go of it: another test process (pytest -n) can be handed the same port in between -- seen once in a few dozen runs --, so a
    launch that dies on the rendezvous address is repeated once on a new port."""
    p = None
    for attempt in range(2):
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
               "--master-addr", "127.0.0.1", "--master-port", str(free_port()), script, *script_args]
        p = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=cwd)
        taken = any(w in (p.stderr + p.stdout) for w in ("Address already in use", "EADDRINUSE", "address already in use"))
        if p.returncode == 0 or not taken:
            break
    return p


class Case:
    """read/generate -> x -> CPU reference y -> reorder -> P*x (solver_test.c:350-376)."""

    def __init__(self, E, O, kind, args, cfg, reorder=True, matrix=None):
        self.E, self.O, self.cfg = E, O, cfg
        m = matrix if matrix is not None else E.Matrix.generate(kind, *args, cfg=cfg)
        self.m = m
        self.n = m.n
        self.x = O.x_glibc(m.n)                                   # solver_test.c:89-92
        self.y_ref = O.spmv_coo(m.n, m.I, m.J, m.V, self.x)       # solver_test.c:102 / 247,254
        self.scale = O.abs_rowsum(m.n, m.I, m.J, m.V, self.x)
        self.nnz = m.nnz
        if reorder:
            m.reorder(cfg)                                        # solver_test.c:370/373
        else:
            m.reorder_list[:] = np.arange(m.n, dtype=np.int32)
        self.perm = m.reorder_list.copy()
        self.xp = E.vector_reorder(self.x, self.perm)             # solver_test.c:376

    def recover(self, yp):
        return self.E.vector_recover(yp, self.perm)               # solver_test.c:383

    def check(self, yp, tol=None):
        y = self.recover(yp)
        bad, worst = self.O.check_tolerance(y, self.y_ref, self.scale, tol or self.O.TOLERANCE)
        return bad, worst


SMALL_CASES = [
    # (name, generator, args) -- sized so the CPU oracle takes well under a second each
    ("fem3d_scrambled", "fem3d", (30000, 3, 22, 22, 13500, 1, 1)),
    ("fem3d_natural", "fem3d", (24000, 3, 20, 20, 0, 0, 2)),
    ("stencil5_noise", "stencil2d", (150, 150, 5, 3000, 1)),
    ("stencil9_noise", "stencil2d", (120, 100, 9, 6000, 3)),
    ("rmat_s14", "rmat", (14, 1 << 17, 1)),
    ("rmat_s12_dense", "rmat", (12, 1 << 18, 5)),
    ("banded_16k", "banded", (1 << 14, 32, 1024)),
    ("kkt3d_12", "kkt3d", (12,)),
]


def fem_plus_rmat(E, cfg, fem_rows=30000, rmat_scale=15, rmat_edges=1 << 19):
    """Block-diagonal [FEM-like | R-MAT]: the FEM partitions' LDS windows pay, the R-MAT partitions' do not --
    the input on which only SOME windows are given up to the panel residual (plan.cpp)."""
    a = E.Matrix.generate("fem3d", fem_rows, 3, 22, 22, 13500, 1, 1, cfg=cfg)
    b = E.Matrix.generate("rmat", rmat_scale, rmat_edges, 5, cfg=cfg)
    rp = np.concatenate([a.row_idx.astype(np.int64), a.nnz + b.row_idx.astype(np.int64)[1:]])
    J = np.concatenate([a.J, b.J + a.n])
    V = np.concatenate([a.V, b.V])
    a.free(), b.free()
    return E.Matrix.from_csr(rp, J, V, cfg)
