"""The multiply in PARTS on one GPU (ehyb_plan_create_host_segs + ehyb_spmv_part, what a multi-GPU rank runs while its
ghost columns arrive chunk by chunk), ehyb_gather, and the argument checks of both."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _case(E, O, kind, args, cfg):
    m = E.Matrix.generate(kind, *args, cfg=cfg)
    n = m.n
    x = O.x_glibc(n)
    y_ref = O.spmv_coo(n, m.I, m.J, m.V, x)
    scale = O.abs_rowsum(n, m.I, m.J, m.V, x)
    m.reorder(cfg)
    return m, x, y_ref, scale


@pytest.mark.parametrize("name,kind,args,kw", [
    ("rmat-panel", "rmat", (17, 1 << 20, 1), dict(er_mode=2, fuse_er=2, er_panel_cols=4096)),
    ("rmat-panel-windows-kept", "rmat", (16, 1 << 19, 2), dict(er_mode=2, fuse_er=2, er_panel_cols=2048, ell_prune=2, lds_doubles=4096)),
    ("rmat-csr", "rmat", (16, 1 << 19, 3), dict(er_mode=1, fuse_er=2)),
    ("fem-csr-residual", "fem3d", (60000, 3, 28, 28, 13500, 1, 1), dict(er_mode=1, fuse_er=2, lds_doubles=1024, window_mode=1)),
], ids=lambda v: v if isinstance(v, str) else None)
def test_parts_add_up_to_the_whole_multiply(E, O, gpu, name, kind, args, kw):
    cfg = E.make_config(direct=2, **kw)          # (65,536 rows would otherwise take the direct shape, which has no parts)
    m, x, y_ref, scale = _case(E, O, kind, args, cfg)
    n = m.n
    perm = m.reorder_list.copy()
    segs = np.array([0, (n // 3) & ~1, (n // 3) & ~1, (2 * n // 3) & ~1, n], dtype=np.int32)   # four segments, one of them empty
    plan = E.Plan(m, cfg, col_segs=segs)
    whole = E.Plan(m, cfg)
    assert plan.col_segs == 4 and whole.col_segs == 1 and np.array_equal(plan.array("col_seg_first"), segs)
    st = plan.stats
    if st["er_partials"]:
        # a panel never straddles a segment boundary; the items of a segment are a run of the item list
        u1, it1, si = plan.array("pb_units1").reshape(-1, 4), plan.array("pb_items1").reshape(-1, 2), plan.array("pb_seg_item")
        assert len(si) == 5 and si[0] == 0 and si[-1] == len(it1) and si[2] == si[1]
        for s in range(4):
            if si[s + 1] > si[s]:
                uu = u1[it1[si[s], 0]:it1[si[s + 1] - 1, 1]]
                assert np.all(uu[:, 0] >= segs[s]) and np.all(uu[:, 0] + uu[:, 1] <= segs[s + 1])
    dx = E.DeviceBuffer(n).upload(E.vector_reorder(x, perm))
    dy, dw = E.DeviceBuffer(n), E.DeviceBuffer(n)
    dy.upload(np.full(n, np.nan))
    # the parts in the order a rank issues them: own columns first (ELL), then segment by segment, the closing pass last
    plan.spmv_part(dx.ptr, dy.ptr, 0, 0, 1, 1)
    plan.spmv_part(dx.ptr, dy.ptr, 0, 1, 3, 0)
    plan.spmv_part(dx.ptr, dy.ptr, 0, 3, 4, 2)
    whole.spmv(dx.ptr, dw.ptr)
    E.host._lib.load().ehyb_dev_sync()
    y_parts, y_whole = E.vector_recover(dy.download(), perm), E.vector_recover(dw.download(), perm)
    for y in (y_parts, y_whole):
        bad, worst = O.check_tolerance(y, y_ref, scale)
        assert bad == 0, (name, worst)
    # one call with every segment and both flags is the whole multiply
    dy.upload(np.full(n, np.nan))
    plan.spmv_part(dx.ptr, dy.ptr, 0, 0, 4, 3)
    E.host._lib.load().ehyb_dev_sync()
    assert O.check_tolerance(E.vector_recover(dy.download(), perm), y_ref, scale)[0] == 0


def test_part_and_segment_arguments_are_checked(E, O, gpu):
    lib = E.host._lib.load()
    cfg = E.make_config(er_mode=2, fuse_er=2, er_panel_cols=2048, direct=2)
    m, x, _, _ = _case(E, O, "rmat", (15, 1 << 18, 1), cfg)
    n = m.n
    h = C.c_void_p()
    for bad in ([0, 101, n], [0, n // 2, n // 4, n], [2, n], [0, n - 2]):     # odd start, descending, not from 0, not to n
        arr = np.asarray(bad, dtype=np.int32)
        rc = lib.ehyb_plan_create_host_segs(C.byref(m.c), 0, n, C.byref(cfg), len(arr) - 1, arr.ctypes.data_as(C.POINTER(C.c_int)), C.byref(h))
        assert rc == 1 and not h.value, bad                                    # EHYB_ERR_ARG
    plan = E.Plan(m, cfg, col_segs=np.array([0, n // 2 & ~1, n], dtype=np.int32))
    dx, dy = E.DeviceBuffer(n).upload(x), E.DeviceBuffer(n)
    for s0, s1 in ((-1, 1), (0, 3), (2, 1)):
        with pytest.raises(E.EhybError) as ei:
            plan.spmv_part(dx.ptr, dy.ptr, 0, s0, s1, 3)
        assert ei.value.code == 1
    # a plan that multiplies in one launch has no parts
    small = E.Matrix.generate("fem3d", 30000, 3, 22, 22, 13500, 1, 1)
    small.reorder()
    direct = E.Plan(small)
    with pytest.raises(E.EhybError) as ei:
        direct.spmv_part(dx.ptr, dy.ptr, 0, 0, 1, 3)
    assert ei.value.code == 8                                                  # EHYB_ERR_STATE


def test_gather_packs_the_send_list(E, gpu):
    lib = E.host._lib.load()
    rng = np.random.default_rng(5)
    n, k = 100000, 37011                         # not a multiple of the 1024 entries a workgroup packs
    src = rng.standard_normal(n)
    idx = rng.integers(0, n, size=k).astype(np.int32)
    ds, dd = E.DeviceBuffer(n).upload(src), E.DeviceBuffer(k)
    di = C.c_void_p()
    assert lib.ehyb_dev_alloc(k * 4, C.byref(di)) == 0 and lib.ehyb_h2d(di, idx.ctypes.data_as(C.c_void_p), k * 4) == 0
    assert lib.ehyb_gather(C.c_void_p(ds.ptr), di, C.c_void_p(dd.ptr), k, None) == 0
    assert lib.ehyb_dev_sync() == 0 and np.array_equal(dd.download(), src[idx])
    assert lib.ehyb_gather(None, None, None, 0, None) == 0 and lib.ehyb_gather(None, di, C.c_void_p(dd.ptr), 5, None) == 1
    lib.ehyb_dev_free(di)
