"""Random small matrices and random plan configurations for the fuzz tests (CPU walk and GPU)."""
import numpy as np
import scipy.sparse as sp


def random_matrix(rng):
    kind = int(rng.integers(0, 5))
    n = int(rng.choice([1, 2, 5, 63, 64, 65, 130, 500, 1500, 3000]))
    dens = float(rng.choice([0.0, 0.001, 0.01, 0.05, 0.3]))
    A = sp.random(n, n, density=dens, random_state=int(rng.integers(1 << 30)), format="csr")
    if kind == 1:                       # symmetric
        A = (A + A.T).tocsr()
    elif kind == 2 and n <= 1000:       # symmetric with dense dof x dof blocks (shared column lists, group sums)
        dof = int(rng.choice([2, 3, 6]))
        B = sp.random(n, n, density=min(dens, 0.02), random_state=int(rng.integers(1 << 30)), format="csr")
        B = (B + B.T + sp.identity(n)).tocsr()
        blk = np.arange(1, dof * dof + 1).reshape(dof, dof) + 0.5
        A = sp.kron(B, blk + blk.T).tocsr()
    elif kind == 3:                     # a dense row, sometimes its column too
        A = A.tolil()
        r = int(rng.integers(0, n))
        A[r, :] = 1.25
        if rng.random() < 0.5:
            A[:, r] = 1.25
        A = A.tocsr()
    elif kind == 4:                     # full diagonal
        A = (A + sp.diags(rng.uniform(1, 2, n))).tocsr()
    A.sort_indices()
    return A


def random_config_kwargs(rng):
    mode = int(rng.choice([1, 2]))
    return dict(window_mode=mode, lds_doubles=int(rng.choice([64, 128, 256, 1024, 2048, 20480])),
                threads=int(rng.choice([256, 512, 1024])), fuse_er=int(rng.choice([0, 1, 2])),
                er_seg_len=int(rng.choice([16, 64, 1024])), items_per_cu=int(rng.choice([0, 1, 2, 4])),
                col_sharing=int(rng.choice([1, 2])), sym_pairs=int(rng.choice([0, 1])) if mode == 2 else 0,
                cap_split=int(rng.choice([1, 2])), hub_rule=int(rng.choice([1, 2])),
                er_mode=int(rng.choice([0, 1, 2])), er_panel_cols=int(rng.choice([256, 1024, 8192, 16384])),
                er_block_rows=int(rng.choice([64, 1000, 8192])), direct=int(rng.choice([0, 0, 1, 2])),
                ell_prune=int(rng.choice([1, 1, 2])),
                # round 3: row order, how pass 1 of the panel form adds up, finds its work and is launched
                partitioner=int(rng.choice([0, 0, 1, 4])), er_sums=int(rng.choice([1, 1, 2])), er_queue=int(rng.choice([0, 1, 1, 2])),
                # (3000 items and more: more than the resident workgroups, so that the per-XCD queues, their walk from the far end and the skip of a
                # panel already staged really run -- round 4)
                xcd_map=int(rng.choice([1, 2])), er_panel_threads=int(rng.choice([0, 512, 1024])), er_units1=int(rng.choice([0, 1, 7, 300, 3000, 4000])),
                er_units2=int(rng.choice([0, 5])), graph_compress=int(rng.choice([0, 1, 2])),
                # where the panel form is built when the plan is created and uploaded in one call (1 host, 2 device)
                symbolic=int(rng.choice([1, 2])),
                # successive multiplies walking the streams in alternating directions (0 by size, 1 always, 2 never)
                ell_alternate=int(rng.choice([0, 1, 1, 2])))


def build(E, O, seed):
    """-> (matrix, plan config, x in original order, oracle y, tolerance scale); the matrix is reordered."""
    rng = np.random.default_rng(seed)
    A = random_matrix(rng)
    kw = random_config_kwargs(rng)
    cfg = E.make_config(**kw)
    m = E.Matrix.from_csr(A.indptr, A.indices, A.data, cfg)
    x = O.x_glibc(m.n)
    if m.nnz:
        y_ref = O.spmv_coo(m.n, m.I, m.J, m.V, x)
        scale = O.abs_rowsum(m.n, m.I, m.J, m.V, x)
        m.reorder(cfg, symmetric=bool(rng.integers(0, 2)) and abs(A - A.T).nnz == 0)
    else:
        y_ref, scale = np.zeros(m.n), np.zeros(m.n)
        m.reorder_list[:] = np.arange(m.n, dtype=np.int32)
        m.c.nParts = 1
        m.part_boundary[:] = [0, m.n]
    return m, cfg, kw, x, y_ref, scale + 1e-300
