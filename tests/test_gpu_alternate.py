"""Successive multiplies of a plan walk their streams in alternating directions (cfg.ell_alternate; ELL slabs and pass 1 of the
panel residual): the direction changes the order of the work, never the result."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _case(E, O, kind, args, cfg):
    m = E.Matrix.generate(kind, *args, cfg=cfg)
    x = O.x_glibc(m.n)
    y_ref = O.spmv_coo(m.n, m.I, m.J, m.V, x)
    scale = O.abs_rowsum(m.n, m.I, m.J, m.V, x)
    m.reorder(cfg)
    return m, x, y_ref, scale


@pytest.mark.parametrize("name,kind,args,kw,deterministic", [
    ("fem-plain", "fem3d", (60000, 3, 28, 28, 13500, 1, 1), dict(lds_doubles=4096), True),
    ("fem-reference-window", "fem3d", (60000, 3, 28, 28, 13500, 1, 1), dict(lds_doubles=1024, window_mode=1), False),
    ("fem-symmetric-pairs", "fem3d", (60000, 3, 28, 28, 13500, 1, 1), dict(lds_doubles=4096, sym_pairs=1), False),
    ("banded-relative-columns", "banded", (1024 * 64, 32, 1024), dict(), True),
    ("rmat-panel", "rmat", (17, 1 << 20, 3), dict(er_mode=2, fuse_er=2, direct=2, er_panel_cols=4096), False),
    ("rmat-panel-units", "rmat", (16, 1 << 19, 5), dict(er_mode=2, fuse_er=2, direct=2, er_units1=7, er_panel_cols=1024), False),
    # the per-XCD work queues (automatic from six items per resident workgroup up): every eighth taken from its far end on the way back,
    # and a workgroup that draws the next item of the panel it has staged streams straight away
    ("rmat-panel-queues", "rmat", (17, 1 << 20, 3), dict(er_mode=2, fuse_er=2, direct=2, er_units1=3000, er_panel_cols=2048, er_queue=1), False),
    ("rmat-panel-queues-wide", "rmat", (17, 1 << 20, 3), dict(er_mode=2, fuse_er=2, direct=2, er_units1=1200, er_queue=1), False),
], ids=lambda v: v if isinstance(v, str) else None)
def test_both_directions_give_the_product(E, O, gpu, name, kind, args, kw, deterministic):
    cfg = E.make_config(ell_alternate=1, **kw)          # (1 = always: these matrices are far smaller than the cache)
    m, x, y_ref, scale = _case(E, O, kind, args, cfg)
    perm = m.reorder_list
    plan = E.Plan(m, cfg)
    dx = E.DeviceBuffer(m.n).upload(E.vector_reorder(x, perm))
    ys = []
    for _ in range(4):                                   # first to last, last to first, and again
        dy = E.DeviceBuffer(m.n)
        dy.upload(np.full(m.n, np.nan))
        plan.spmv(dx.ptr, dy.ptr)
        E.host._lib.load().ehyb_dev_sync()
        ys.append(dy.download())
        bad, worst = O.check_tolerance(E.vector_recover(ys[-1], perm), y_ref, scale)
        assert bad == 0, (name, len(ys), worst)
    if deterministic:                                    # a row's sum does not depend on the order its slab is taken in
        assert np.array_equal(ys[0], ys[1]) and np.array_equal(ys[0], ys[2])
    never = E.Plan(m, E.make_config(ell_alternate=2, **kw))
    dy = E.DeviceBuffer(m.n)
    never.spmv(dx.ptr, dy.ptr)
    E.host._lib.load().ehyb_dev_sync()
    if deterministic:
        assert np.array_equal(dy.download(), ys[1])


def test_two_host_threads_share_one_plan(E, O, gpu):
    """The walk direction is per-plan state flipped atomically by every launch: two host threads multiplying with ONE plan on their
    own streams and vectors (an ELL-only plan, plain storage) get bit-identical results -- a row's sum does not depend on the
    direction its slab was taken in -- and together they draw every direction exactly once per launch."""
    import threading

    cfg = E.make_config(ell_alternate=1, lds_doubles=4096)
    m, x, y_ref, scale = _case(E, O, "fem3d", (60000, 3, 28, 28, 13500, 1, 1), cfg)
    perm = m.reorder_list
    plan = E.Plan(m, cfg)
    assert plan.stats["er_partials"] == 0
    dx = E.DeviceBuffer(m.n).upload(E.vector_reorder(x, perm))
    single = E.DeviceBuffer(m.n)
    plan.spmv(dx.ptr, single.ptr)
    E.host._lib.load().ehyb_dev_sync()
    y_one = single.download()
    assert O.check_tolerance(E.vector_recover(y_one, perm), y_ref, scale)[0] == 0
    results, errors = {}, []

    def worker(tid):
        try:
            st, dy = E.Stream(), E.DeviceBuffer(m.n)
            outs = []
            for it in range(40):
                dy.upload(np.full(m.n, np.nan))
                plan.spmv(dx.ptr, dy.ptr, st.ptr)
                st.sync()
                if it % 8 == 7:
                    outs.append(dy.download())
            results[tid] = outs
            st.destroy()
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    ts = [threading.Thread(target=worker, args=(t,)) for t in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors
    for tid in (0, 1):
        assert len(results[tid]) == 5 and all(np.array_equal(y, y_one) for y in results[tid])


@pytest.mark.parametrize("kw,deterministic", [(dict(lds_doubles=4096), True), (dict(lds_doubles=4096, sym_pairs=1), False),
                                              (dict(er_mode=2, fuse_er=2, direct=2, er_panel_cols=4096), False)], ids=["plain", "sym", "panel"])
def test_explicit_walk_and_captured_graphs_keep_the_product(E, O, gpu, kw, deterministic):
    """ehyb_spmv_walk states the direction per call; ehyb_spmv_graph_create captures a run of multiplies with the directions
    alternating explicitly (two executables for an odd run, replayed in turn): the product every time."""
    gen = ("rmat", (17, 1 << 20, 3)) if "er_mode" in kw else ("fem3d", (60000, 3, 28, 28, 13500, 1, 1))
    cfg = E.make_config(**kw)                              # ell_alternate automatic: these plans do not alternate by themselves
    m, x, y_ref, scale = _case(E, O, gen[0], gen[1], cfg)
    perm = m.reorder_list
    plan = E.Plan(m, cfg)
    dx, dy = E.DeviceBuffer(m.n).upload(E.vector_reorder(x, perm)), E.DeviceBuffer(m.n)
    sync = E.host._lib.load().ehyb_dev_sync
    ys = []
    for walk in (0, 1, None):
        dy.upload(np.full(m.n, np.nan))
        plan.spmv(dx.ptr, dy.ptr, walk=walk)
        sync()
        ys.append(dy.download())
        assert O.check_tolerance(E.vector_recover(ys[-1], perm), y_ref, scale)[0] == 0, walk
    if deterministic:
        assert np.array_equal(ys[0], ys[1]) and np.array_equal(ys[0], ys[2])
    st = E.Stream()
    for count in (1, 4):
        g = plan.graph(dx.ptr, dy.ptr, count)
        for launch in range(3):
            dy.upload(np.full(m.n, np.nan))
            g.launch(st.ptr)
            st.sync()
            y = dy.download()
            assert O.check_tolerance(E.vector_recover(y, perm), y_ref, scale)[0] == 0, (count, launch)
            if deterministic:
                assert np.array_equal(y, ys[0])
        g.destroy()
    st.destroy()
    with pytest.raises(E.EhybError):
        plan.spmv(dx.ptr, dy.ptr, walk=2)
