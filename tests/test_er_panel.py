"""Panel form of a large residual (csrc/er_panel.cpp; kernels ehyb_pb_scale_kernel / ehyb_pb_reduce_kernel):
the host builder checked without a GPU by the oracle's walk of the arrays (oracle.walk_panel_residual
indexes them as the kernels do and checks the invariants they rely on), against the CSR segments of
the same residual and against the reference CPU product."""
import numpy as np
import pytest

from util import Case, fem_plus_rmat

CASES = [
    ("rmat_s14", "rmat", (14, 1 << 17, 1), dict(lds_doubles=512, er_panel_cols=512, er_block_rows=300)),
    ("rmat_s12_dense", "rmat", (12, 1 << 18, 5), dict(lds_doubles=256, er_panel_cols=256, er_block_rows=64)),
    ("kkt_contiguous", "kkt3d", (12,), dict(lds_doubles=512, partitioner=1)),
    ("rmat_s16_defaults", "rmat", (16, 1 << 19, 3), dict(direct=2)),
    ("fem_reference_window", "fem3d", (30000, 3, 22, 22, 13500, 1, 1), dict(window_mode=1, lds_doubles=1024, er_panel_cols=1024)),
    ("rmat_sym_storage", "rmat", (14, 1 << 17, 4), dict(lds_doubles=512, sym_pairs=1, er_panel_cols=256)),
]


@pytest.mark.parametrize("name,kind,args,kw", CASES, ids=[c[0] for c in CASES])
def test_panel_form_walks_to_the_reference_product(E, O, name, kind, args, kw):
    cfg = E.make_config(er_mode=2, fuse_er=2, **kw)
    c = Case(E, O, kind, args, cfg)
    plan = E.Plan(c.m, cfg, upload=False)
    st = plan.stats
    assert st["nnz_er"] > 0 and 0 < st["er_partials"] <= st["nnz_er"]
    y, written = O.walk_plan(plan, c.xp)          # asserts panel form == CSR segments inside
    assert written[:c.n].min() == 1
    bad, worst = c.check(y)
    assert bad == 0, f"worst {worst:.3e}"
    # pass-1 units stay inside one panel, pass-2 row blocks inside the limit
    u1 = plan.array("pb_units1").reshape(-1, 4)
    u2 = plan.array("pb_units2").reshape(-1, 4)
    assert np.all(u1[:, 1] <= cfg.er_panel_cols) and np.all(u1[:, 0] % cfg.er_panel_cols == 0)
    assert np.all(np.abs(u2[:, 3]) <= cfg.er_block_rows)   # (negative: a block that assigns y, pb_assign)
    # bytes: both passes streamed -- entries at 10 B (value, column word with the two slot flags) plus the jump
    # list, partials at 18 B, plus panels and y; the jump list holds at most one slot per partial
    er_bytes = st["bytes_format"] - st["bytes_format_ell"]
    jumps = len(plan.array("pb_jump"))
    assert jumps <= st["er_partials"] + len(plan.array("pb_chunk"))
    assert er_bytes >= 10 * st["nnz_er"] + 18 * st["er_partials"] + 4 * jumps


def test_mode_selection(E, O):
    """er_mode 1 = CSR segments, 2 = panel form, 0 = panel form only for a large residual (2^21 entries up)
    without locality; a tiny residual rides inside the ELL launch whatever er_mode says."""
    cfg1 = E.make_config(er_mode=1, lds_doubles=512)
    c = Case(E, O, "rmat", (14, 1 << 17, 1), cfg1)
    assert E.Plan(c.m, cfg1, upload=False).stats["er_partials"] == 0
    assert E.Plan(c.m, E.make_config(er_mode=0, lds_doubles=512), upload=False).stats["er_partials"] == 0   # 64 k entries: CSR
    assert E.Plan(c.m, E.make_config(er_mode=2, lds_doubles=512), upload=False).stats["er_partials"] > 0
    cfg = E.make_config(er_mode=2, direct=2)
    f = Case(E, O, "fem3d", (30000, 3, 22, 22, 13500, 1, 1), cfg)
    st = E.Plan(f.m, cfg, upload=False).stats
    assert st["er_partials"] == 0 and (st["nnz_er"] == 0 or st["er_inline"] > 0)


def test_hub_row_collapses_to_few_partials(E, O):
    """A row with tens of thousands of residual entries leaves at most one partial per 64-entry chunk it
    touches -- far fewer than its entries (what keeps the second pass balanced on power-law inputs)."""
    cfg = E.make_config(er_mode=2, lds_doubles=256, er_panel_cols=4096)
    c = Case(E, O, "rmat", (15, 1 << 20, 7), cfg)
    plan = E.Plan(c.m, cfg, upload=False)
    st = plan.stats
    assert st["max_row"] > 4000 and st["er_partials"] < 0.5 * st["nnz_er"]
    y, _ = O.walk_plan(plan, c.xp)
    assert c.check(y)[0] == 0


def test_plan_cache_round_trip_with_panel_form(E, O, tmp_path):
    cfg = E.make_config(er_mode=2, lds_doubles=512, er_panel_cols=512)
    c = Case(E, O, "rmat", (13, 1 << 16, 2), cfg)
    plan = E.Plan(c.m, cfg, upload=False)
    path = tmp_path / "p.ehyb"
    plan.save(path, c.perm, key=7)
    loaded, perm = E.Plan.load(path, key=7, upload=False)
    assert loaded.stats == plan.stats and np.array_equal(perm, c.perm)
    for name in ("pb_val", "pb_col", "pb_dst", "pb_units1", "pb_items1", "pb_row", "pb_units2"):
        assert np.array_equal(loaded.array(name), plan.array(name)), name
    y, _ = O.walk_plan(loaded, c.xp)
    assert c.check(y)[0] == 0
    # a slot beyond the partial buffer must not reach the GPU
    data = bytearray(path.read_bytes())
    dst = plan.array("pb_dst")
    needle = dst[:8].tobytes()
    off = bytes(data).rindex(needle)
    data[off:off + 4] = (plan.stats["er_partials"] + 5).to_bytes(4, "little")
    bad = tmp_path / "bad.ehyb"
    bad.write_bytes(bytes(data))
    with pytest.raises(E.EhybError) as e:
        E.Plan.load(bad, upload=False)
    assert e.value.code == 6


def test_windows_that_do_not_pay_go_to_the_residual(E, O):
    """With the residual in panel form a partition whose LDS window costs more (padding, halo gathers)
    than the panel form would for its entries is given to the residual whole: no window, no halo,
    zero-width slabs (cfg.ell_prune; plan.cpp windows_that_do_not_pay)."""
    kw = dict(er_mode=2, lds_doubles=1024, er_panel_cols=1024)
    c = Case(E, O, "rmat", (15, 1 << 19, 5), E.make_config(**kw))
    kept = E.Plan(c.m, E.make_config(ell_prune=2, **kw), upload=False)
    pruned = E.Plan(c.m, E.make_config(**kw), upload=False)
    a, b = kept.stats, pruned.stats
    assert b["nnz_ell"] < a["nnz_ell"] and b["nnz_er"] > a["nnz_er"] and b["nnz_ell"] + b["nnz_er"] == c.nnz
    assert b["halo_cols"] < a["halo_cols"]
    wl = pruned.array("win_len")
    assert (wl == 0).any() and (wl > 0).any()
    # a partition without a window stages nothing and its slabs are empty
    segs = pruned.array("segs").reshape(-1, 8)
    meta = pruned.array("slab_meta").reshape(-1, 4)
    for p_, s0, s1, hn, ps, pe, w, hb in segs:
        if w == 0:
            assert hn == 0 and not (meta[s0:s1, 3] >> 16).any()
    y, written = O.walk_plan(pruned, c.xp)
    assert written[:c.n].min() == 1 and c.check(y)[0] == 0


@pytest.mark.parametrize("mixed", [True, False], ids=["some_windows_kept", "no_window_kept"])
def test_rows_of_partitions_without_a_window_are_assigned_by_pass_2(E, O, mixed):
    """A partition whose window does not pay goes to the panel residual whole (plan.cpp); where less than a quarter
    of the entries would be left in windows, all of them go.  The rows of such partitions get y from pass 2 alone:
    row blocks that ASSIGN (rows stored negative), present also where no partial arrives; the ELL launch has no
    segment for them and its work items carry the kept windows only.  No block mixes both kinds of rows."""
    cfg = E.make_config(partitioner=E.EHYB_PART_CONTIGUOUS, er_mode=2, lds_doubles=4096)
    if mixed:
        c = Case(E, O, None, None, cfg, matrix=fem_plus_rmat(E, cfg))
    else:
        c = Case(E, O, "rmat", (18, 1 << 21, 1), cfg)   # a fifth of the entries in windows that pay: all given up
    plan = E.Plan(c.m, cfg, upload=False)
    st = plan.stats
    wl, pb = plan.array("win_len"), plan.array("part_boundary")
    u2 = plan.array("pb_units2").reshape(-1, 4)
    windowless = wl == 0
    assert windowless.any() and (not windowless.all()) == mixed and st["er_partials"] > 0
    row_kind = np.repeat(windowless, np.diff(pb))                     # per row: its partition has no window
    for first, end, r0, rows in u2:
        kinds = row_kind[r0:r0 + abs(rows)]
        assert np.all(kinds == (rows < 0)), "a row block mixes assigned and accumulated rows"
    covered = np.zeros(c.n, dtype=int)
    for first, end, r0, rows in u2[u2[:, 3] < 0]:
        covered[r0:r0 - rows] += 1
    assert np.array_equal(covered == 1, row_kind), "every row without a window belongs to exactly one assigning block"
    y, written = O.walk_plan(plan, c.xp)
    assert written[:c.n].min() == 1 and written[:c.n].max() == 1
    assert c.check(y)[0] == 0
    segs = plan.array("segs").reshape(-1, 8)
    if mixed:
        assert st["n_items"] > 1 and len(segs) > 0 and np.all(wl[segs[:, 0]] > 0) and st["nnz_ell"] * 4 >= st["nnz"]
    else:
        assert st["nnz_ell"] == 0 and st["n_items"] == 1 and len(segs) == 0


def test_column_segments_cut_the_panels_and_the_items(E, O):
    """ehyb_plan_create_host_segs (multi-GPU: the ghost columns arrive segment by segment): a panel never straddles a
    segment boundary, the pass-1 items of a segment are a run of the item list, every entry is still multiplied once
    (oracle walk), and malformed segment lists are refused."""
    import ctypes as C

    cfg = E.make_config(er_mode=2, fuse_er=2, er_panel_cols=1024, direct=2, partitioner=E.EHYB_PART_DEGREE)
    c = Case(E, O, "rmat", (15, 1 << 18, 4), cfg)
    n = c.n
    segs = np.array([0, 4098, 4098, 20000, n], dtype=np.int32)      # an empty segment among them; starts even
    plan = E.Plan(c.m, cfg, upload=False, col_segs=segs)
    assert plan.col_segs == 4 and np.array_equal(plan.array("col_seg_first"), segs)
    u1 = plan.array("pb_units1").reshape(-1, 4)
    it1 = plan.array("pb_items1").reshape(-1, 2)
    si = plan.array("pb_seg_item")
    assert len(si) == 5 and si[0] == 0 and si[-1] == len(it1) and si[1] < si[-1] and si[2] == si[1]
    for s in range(4):
        if si[s + 1] > si[s]:
            uu = u1[it1[si[s], 0]:it1[si[s + 1] - 1, 1]]
            assert np.all(uu[:, 0] >= segs[s]) and np.all(uu[:, 0] + uu[:, 1] <= segs[s + 1])
            assert np.all((uu[:, 0] - segs[s]) % 1024 == 0)           # panels restart at the segment's first column
    y, written = O.walk_plan(plan, c.xp)
    assert written[:n].min() == 1 and c.check(y)[0] == 0
    # the same plan without segments holds the same entries in (possibly) fewer panels
    whole = E.Plan(c.m, cfg, upload=False)
    assert whole.stats["nnz_er"] == plan.stats["nnz_er"] and whole.stats["er_partials"] <= plan.stats["er_partials"]
    lib = E.host._lib.load()
    h = C.c_void_p()
    for bad in ([0, 4097, n], [0, 9000, 4096, n], [2, n], [0, n - 2]):
        arr = np.asarray(bad, dtype=np.int32)
        assert lib.ehyb_plan_create_host_segs(C.byref(c.m.c), 0, n, C.byref(cfg), len(arr) - 1, arr.ctypes.data_as(C.POINTER(C.c_int)), C.byref(h)) == 1
        assert not h.value
