"""The panel form of the residual built ON THE DEVICE (ehyb_plan_create / ehyb_plan_create_segs with cfg.symbolic = 2,
csrc/er_panel_dev.hip) against the host builder (csrc/er_panel.cpp): every array of the form the same,
entry for entry; the multiply right against the oracle; the numeric refill, the plan cache and the multiply in parts
working on such a plan."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

PANEL_ARRAYS = ["pb_val", "pb_col", "pb_dst", "pb_colf", "pb_chunk", "pb_jump", "pb_row", "pb_units1", "pb_items1", "pb_units2", "pb_seg_item",
                "pb_src", "col_seg_first"]
# everything the device route leaves to the host builder must be untouched by it as well
HOST_ARRAYS = ["part_boundary", "win_len", "halo_ptr", "halo_cols", "slab_pair_ptr", "slab_row", "slab_part", "ell_val", "ell_col",
               "slab_col_ptr", "lane_group", "slab_meta", "segs", "ell_src"]


def _case(E, O, kind, args, cfg):
    m = E.Matrix.generate(kind, *args, cfg=cfg)
    x = O.x_glibc(m.n)
    y_ref = O.spmv_coo(m.n, m.I, m.J, m.V, x)
    scale = O.abs_rowsum(m.n, m.I, m.J, m.V, x)
    m.reorder(cfg)
    return m, x, y_ref, scale


def _both(E, m, kw, **plan_kw):
    host = E.Plan(m, E.make_config(symbolic=1, **kw), **plan_kw)
    dev = E.Plan(m, E.make_config(symbolic=2, **kw), **plan_kw)
    return host, dev


def _same_form(host, dev):
    sh, sd = host.stats, dev.stats
    assert sd["er_partials"] > 0 and sd["er_segments"] == 0 and sh["er_segments"] > 0       # the device route builds no CSR segments
    for k in sh:
        if k != "er_segments":
            assert sh[k] == sd[k], (k, sh[k], sd[k])
    for name in PANEL_ARRAYS + HOST_ARRAYS:
        a, b = host.array(name), dev.array(name)
        assert a.shape == b.shape and np.array_equal(a, b), (name, a.shape, b.shape)
    items_h, items_d = host.array("items").reshape(-1, 8), dev.array("items").reshape(-1, 8)
    assert np.array_equal(items_h[:, :4], items_d[:, :4])                                    # (words 4..7: the CSR segments of the item)


CASES = [
    # every partition given up (the sampled pre-screen of plan.cpp): the device reads the caller's arrays
    ("rmat-all-residual", "rmat", (18, 1 << 22, 1), dict()),
    ("rmat-all-residual-slot-map", "rmat", (18, 1 << 22, 2), dict(value_map=1, er_panel_cols=4096)),
    # the panel form asked for, windows kept: the residual is a part of the entries (row-order copies go up)
    ("rmat-windows-kept", "rmat", (16, 1 << 19, 3), dict(er_mode=2, fuse_er=2, ell_prune=2, lds_doubles=4096, er_panel_cols=2048, direct=2)),
    ("rmat-windows-kept-slot-map", "rmat", (16, 1 << 19, 4), dict(er_mode=2, fuse_er=2, ell_prune=2, lds_doubles=2048, value_map=1, direct=2)),
    ("fem-small-window", "fem3d", (60000, 3, 28, 28, 13500, 1, 1), dict(er_mode=2, fuse_er=2, lds_doubles=1024, window_mode=1, er_block_rows=1000)),
    ("rmat-two-builds", "rmat", (17, 1 << 20, 5), dict(er_mode=2, fuse_er=2, er_units1=7, er_units2=5, direct=2)),
]


@pytest.mark.parametrize("name,kind,args,kw", CASES, ids=[c[0] for c in CASES])
def test_device_built_panel_form_equals_the_host_builders(E, O, gpu, name, kind, args, kw):
    m, x, y_ref, scale = _case(E, O, kind, args, E.make_config(**kw))
    host, dev = _both(E, m, kw)
    _same_form(host, dev)
    perm = m.reorder_list
    for plan in (host, dev):
        y = E.vector_recover(plan.spmv_host(E.vector_reorder(x, perm)), perm)
        bad, worst = O.check_tolerance(y, y_ref, scale)
        assert bad == 0, (name, worst)


def test_device_built_plan_in_column_segments(E, O, gpu):
    """ehyb_plan_create_segs: panels restart at every segment; the multiply in parts adds up."""
    kw = dict(er_mode=2, fuse_er=2, er_panel_cols=4096, direct=2)
    m, x, y_ref, scale = _case(E, O, "rmat", (17, 1 << 20, 6), E.make_config(**kw))
    n = m.n
    segs = np.array([0, (n // 3) & ~1, (n // 3) & ~1, (2 * n // 3) & ~1, n], dtype=np.int32)
    host, dev = _both(E, m, kw, col_segs=segs)
    _same_form(host, dev)
    perm = m.reorder_list
    dx = E.DeviceBuffer(n).upload(E.vector_reorder(x, perm))
    dy = E.DeviceBuffer(n)
    dy.upload(np.full(n, np.nan))
    dev.spmv_part(dx.ptr, dy.ptr, 0, 0, 1, 1)
    dev.spmv_part(dx.ptr, dy.ptr, 0, 1, 3, 0)
    dev.spmv_part(dx.ptr, dy.ptr, 0, 3, 4, 2)
    E.host._lib.load().ehyb_dev_sync()
    bad, worst = O.check_tolerance(E.vector_recover(dy.download(), perm), y_ref, scale)
    assert bad == 0, worst


def test_refill_and_cache_of_a_device_built_plan(E, O, gpu, tmp_path):
    """ehyb_plan_set_values through the slot map the device kept; ehyb_plan_save fetches the streams, the loaded plan multiplies."""
    kw = dict(value_map=1)
    m, x, y_ref, scale = _case(E, O, "rmat", (18, 1 << 22, 7), E.make_config(**kw))
    perm = m.reorder_list.copy()
    dev = E.Plan(m, E.make_config(symbolic=2, **kw))
    assert dev.stats["er_partials"] > 0 and dev.stats["er_segments"] == 0
    xp = E.vector_reorder(x, perm)
    # saved before any array was asked for
    path = tmp_path / "dev.plan"
    dev.save(path, perm)
    loaded, lperm = E.Plan.load(path)
    assert np.array_equal(lperm, perm)
    y = E.vector_recover(loaded.spmv_host(xp), perm)
    bad, worst = O.check_tolerance(y, y_ref, scale)
    assert bad == 0, worst
    # new values on the same pattern
    rng = np.random.default_rng(5)
    v2 = rng.uniform(-1.0, 1.0, m.nnz)
    dev.set_values(v2)
    y2 = dev.spmv_host(xp)                       # (compared in the plan's own numbering: m is the permuted matrix)
    bad, worst = O.check_tolerance(y2, O.spmv_coo(m.n, m.I, m.J, v2, xp), O.abs_rowsum(m.n, m.I, m.J, v2, xp) + 1e-300)
    assert bad == 0, worst


def test_rows_that_are_not_in_column_order(E, O, gpu):
    """Rows that do not arrive in column order (what the reorder step leaves: it keeps a row's old order), one of them long
    enough for dozens of CSR segments: both builders deal a row out in column order, as a whole."""
    import scipy.sparse as sp

    rng = np.random.default_rng(11)
    n = 40000
    A = sp.random(n, n, density=2.5e-4, random_state=3, format="csr")
    A = (A + sp.csr_matrix((np.ones(3000), (np.zeros(3000, dtype=int), rng.choice(n, 3000, replace=False))), shape=(n, n))).tocsr()
    A.sort_indices()
    indptr, indices, data = A.indptr.copy(), A.indices.copy(), A.data.copy()
    for r in range(n):                           # shuffle every row's entries
        a, b = indptr[r], indptr[r + 1]
        p = rng.permutation(b - a)
        indices[a:b], data[a:b] = indices[a:b][p], data[a:b][p]
    kw = dict(er_mode=2, fuse_er=2, direct=2, lds_doubles=1024, er_panel_cols=1024, er_seg_len=64)
    cfg = E.make_config(**kw)
    m = E.Matrix.from_csr(indptr, indices, data, cfg)
    x = O.x_glibc(n)
    y_ref = O.spmv_coo(n, m.I, m.J, m.V, x)
    scale = O.abs_rowsum(n, m.I, m.J, m.V, x)
    m.reorder_list[:] = np.arange(n, dtype=np.int32)      # no reorder step: the rows stay shuffled
    m.c.nParts = 1
    m.part_boundary[:2] = [0, n]
    host, dev = _both(E, m, kw)
    _same_form(host, dev)
    for plan in (host, dev):
        bad, worst = O.check_tolerance(plan.spmv_host(x), y_ref, scale)
        assert bad == 0, worst


def test_device_builder_gives_its_memory_back(E, O, gpu):
    """Gigabytes of temporaries for a large matrix: after a plan is built on the device and destroyed again, the device has what
    it had before (dozens of temporaries and seven arrays that stay: none may be left behind)."""
    import ctypes as C

    lib = E.host._lib.load()

    def free_bytes():
        f, t = C.c_size_t(), C.c_size_t()
        lib.ehyb_dev_sync()
        assert lib.ehyb_dev_mem_info(C.byref(f), C.byref(t)) == 0
        return f.value

    kw = dict(er_mode=2, fuse_er=2, direct=2, value_map=1)
    cfg = E.make_config(symbolic=2, **kw)
    m = E.Matrix.generate("rmat", 17, 1 << 20, 9, cfg=cfg)
    m.reorder(cfg)
    E.Plan(m, cfg).destroy()                     # first use: kernels loaded
    free0 = free_bytes()
    for _ in range(6):
        plan = E.Plan(m, cfg)
        assert plan.stats["er_partials"] > 0 and plan.stats["er_segments"] == 0
        plan.array("pb_dst")                     # the streams fetched to the host as well
        plan.destroy()
    free1 = free_bytes()
    assert free0 - free1 < (32 << 20), (free0, free1)
