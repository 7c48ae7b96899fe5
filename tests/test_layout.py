"""Host layout builder (the COO2EHYB role): invariants, reference-rule agreement, edge cases.
CPU only: the layout is walked with oracle.walk_plan, which indexes the arrays exactly as the
HIP kernels do."""
import numpy as np
import pytest
import scipy.sparse as sp

from oracle import ehyb_ref_layout as R
from util import SMALL_CASES, Case


def _walk_ok(E, O, c, plan):
    yp, written = O.walk_plan(plan, c.xp)
    assert (written[plan.rows[0]:plan.rows[1]] == 1).all(), "every row is written exactly once by the ELL phase"
    bad, worst = c.check(yp)
    assert bad == 0, f"worst {worst:.3e}"


@pytest.mark.parametrize("name,kind,args", SMALL_CASES, ids=[c[0] for c in SMALL_CASES])
@pytest.mark.parametrize("mode", [1, 2], ids=["refwindow", "halo"])
def test_walk_equals_oracle(E, O, name, kind, args, mode):
    cfg = E.make_config(window_mode=mode, lds_doubles=2048)
    c = Case(E, O, kind, args, cfg)
    plan = E.Plan(c.m, cfg, upload=False)
    _walk_ok(E, O, c, plan)


@pytest.mark.parametrize("fuse", [0, 1, 2], ids=["auto", "inline", "own-launch"])
def test_inline_residual_form(E, O, fuse):
    """A residual the ELL launch carries itself: its entries sit behind the slabs' ELL pairs with
    global columns (walk_plan checks them against the CSR segments, which stay for the two-phase
    call).  fuse_er=1 forces the form, 2 forbids it, 0 takes it for residuals under 0.2 %."""
    cfg = E.make_config(lds_doubles=1024, fuse_er=fuse, cap_split=2)
    c = Case(E, O, "fem3d", (6000, 3, 12, 12, 20000, 1, 5), cfg)
    plan = E.Plan(c.m, cfg, upload=False)
    st = plan.stats
    assert st["nnz_er"] > 0, "the case must leave a residual"
    meta = plan.array("slab_meta").astype(np.int64).reshape(-1, 4)
    ner = (meta[:, 3] >> 8) & 0xFF
    small = st["nnz_er"] * 500 < st["nnz"]
    if fuse == 1 or (fuse == 0 and small):
        assert st["er_inline"] == ner.sum() * 128 > 0
        # a slab's inline pairs hold its longest residual row
        er_len = np.zeros(c.n, dtype=np.int64)
        np.add.at(er_len, plan.array("er_seg_row") & 0x7FFFFFFF, np.diff(plan.array("er_seg_ptr")))
        srow = plan.array("slab_row")
        for s in np.flatnonzero(ner):
            assert ner[s] == (er_len[srow[s]:srow[s] + 64].max() + 1) // 2
    else:
        assert st["er_inline"] == 0 and not ner.any()
    assert st["size_block_ell"] == st["nnz_ell"] + st["ell_padding"]
    assert len(plan.array("ell_val")) == st["size_block_ell"] + st["er_inline"]
    _walk_ok(E, O, c, plan)


@pytest.mark.parametrize("mode", [1, 2])
def test_structural_invariants(E, O, mode):
    """The reference's exit()-style self-checks (convert.c:122-125,226-263,287-303) as assertions:
    (i) every entry lands in exactly one of ELL / residual, (ii) window-local columns stay inside
    the window, (iii) no row exceeds its slab width, (iv) residual rows map back uniquely."""
    cfg = E.make_config(window_mode=mode, lds_doubles=1024, er_seg_len=64)
    c = Case(E, O, "rmat", (13, 1 << 16, 2), cfg)
    plan = E.Plan(c.m, cfg, upload=False)
    st = plan.stats
    A = c.m.to_scipy()
    assert st["nnz_ell"] + st["nnz_er"] == c.nnz == st["nnz"]
    assert st["size_block_ell"] == st["nnz_ell"] + st["ell_padding"]
    pb, wl = plan.array("part_boundary"), plan.array("win_len")
    hp, hc = plan.array("halo_ptr"), plan.array("halo_cols")
    spp, srow, spart = plan.array("slab_pair_ptr").astype(np.int64), plan.array("slab_row"), plan.array("slab_part")
    ev, ew = plan.array("ell_val"), plan.array("ell_col").astype(np.int64)
    scp, lg = plan.array("slab_col_ptr").astype(np.int64), plan.array("lane_group").astype(np.int64).reshape(-1, 64)
    assert pb[0] == 0 and pb[-1] == c.n and np.all(np.diff(pb) > 0)
    assert spp[-1] * 128 == st["size_block_ell"] == len(ev)
    assert scp[-1] == st["col_words"] == len(ew) and st["col_words"] <= st["size_block_ell"] // 2
    meta = plan.array("slab_meta").astype(np.int64).reshape(-1, 4)
    rp, J, V = c.m.row_idx, c.m.J, c.m.V
    er_len = np.zeros(c.n, dtype=np.int64)
    np.add.at(er_len, plan.array("er_seg_row") & 0x7FFFFFFF, np.diff(plan.array("er_seg_ptr")))
    seen, moved_rows = 0, 0
    for s in range(len(srow)):
        p = spart[s]
        ps, pe = pb[p], pb[p + 1]
        base = ps & ~1
        wsize = (ps & 1) + wl[p] + (hp[p + 1] - hp[p])
        assert wsize <= cfg.lds_doubles
        npairs = spp[s + 1] - spp[s]
        G = int(meta[s, 3] & 0x3F) + 1
        # record word 3 = pairs << 16 | inline residual pairs << 8 | groups - 1 (this residual is not inline)
        assert scp[s + 1] - scp[s] == npairs * G and meta[s, 3] >> 16 == npairs and (meta[s, 3] >> 8) & 0xFF == 0
        assert lg[s].max() < G and lg[s][0] == 0 and np.all(np.diff(lg[s]) >= 0) and np.all(np.diff(lg[s]) <= 1)
        words = ew[scp[s]:scp[s + 1]].reshape(npairs, G)[:, lg[s]] if npairs else np.zeros((0, 64), dtype=np.int64)
        cols = np.stack([words & 0xFFFF, words >> 16], axis=2)   # [pair][lane][2] as the kernel decodes it
        vals = ev[spp[s] * 128:spp[s + 1] * 128].reshape(-1, 64, 2)
        assert cols.size == 0 or cols.max() < wsize                              # (ii)
        halo = hc[hp[p]:hp[p + 1]]
        assert np.all(np.diff(halo) > 0) and not np.any((halo >= ps) & (halo < pe))
        for lane in range(64):
            r = srow[s] + lane
            if r >= pe:
                assert not vals[:, lane, :].any()                                # padding lanes are zero
                continue
            rc, rv = J[rp[r]:rp[r + 1]], V[rp[r]:rp[r + 1]]
            in_own = (rc >= ps) & (rc < ps + wl[p])
            in_halo = np.isin(rc, halo)
            k = int(np.count_nonzero(in_own | in_halo))
            if er_len[r] == len(rc) and k > 0:
                # hub row moved to the residual as a whole (it would have padded its slab)
                assert not vals[:, lane, :].any()
                moved_rows += 1
                continue
            assert er_len[r] == len(rc) - k
            assert k <= 2 * cols.shape[0]                                         # (iii)
            flat_c, flat_v = cols[:, lane, :].reshape(-1), vals[:, lane, :].reshape(-1)
            # entries keep the row's storage order; window-local ids decode to the global column
            dec = np.where(flat_c[:k] < (ps & 1) + wl[p], flat_c[:k] + base, 0)
            hal = flat_c[:k] >= (ps & 1) + wl[p]
            dec[hal] = halo[flat_c[:k][hal] - (ps & 1) - wl[p]]
            assert np.array_equal(dec, rc[in_own | in_halo]) and np.array_equal(flat_v[:k], rv[in_own | in_halo])
            assert not flat_v[k:].any()                                          # zero padding (value 0.0)
            seen += k
    assert seen == st["nnz_ell"]                                                  # (i)
    seg_ptr, seg_row = plan.array("er_seg_ptr"), plan.array("er_seg_row")
    assert np.diff(seg_ptr).max() <= cfg.er_seg_len and np.diff(seg_ptr).min() >= 1
    rows = seg_row & 0x7FFFFFFF
    uniq, counts = np.unique(rows, return_counts=True)
    split = set(uniq[counts > 1].tolist())
    assert split == set(rows[seg_row < 0].tolist())                               # (iv) + split flag
    assert len(uniq) == st["rows_er"]
    bins = plan.array("er_bins")
    assert bins[0] == 0 and bins[3] == len(seg_row)
    items = plan.array("items").reshape(-1, 8)
    segs = plan.array("segs").reshape(-1, 8)
    covered = np.zeros(len(srow), dtype=int)
    nxt = 0
    cost = []
    for g0, g1, s0, s1, e0, e64, e16, e1 in items:
        assert s0 < s1 and g0 < g1 and segs[g0, 1] == s0 and segs[g1 - 1, 2] == s1
        for p, a, b, hn, ps, pe, wlen, hb in segs[g0:g1]:
            assert np.all(spart[a:b] == p) and (ps, pe) == (pb[p], pb[p + 1])
            covered[a:b] += 1
        assert e0 == nxt and e0 <= e64 <= e16 <= e1                              # residual segments grouped by item
        lens = np.diff(seg_ptr[e0:e1 + 1])
        assert np.all(np.diff(lens) <= 0), "longest first inside an item"
        nxt = e1
        cost.append(int(spp[s1] - spp[s0]))
    assert nxt == len(seg_row)
    assert np.all(covered == 1), "work items tile the slabs exactly once"
    assert len(items) <= cfg.items_per_cu * 256, "the grid never exceeds the resident-slot budget"
    del A


def test_shared_column_lists(E, O):
    """Rows with the column list of the row above store no column words of their own: a 3-dof
    finite-element matrix needs about a third of the index bytes; switching it off gives one
    list per lane; both walk to the same y."""
    kw = dict(window_mode=2, lds_doubles=2048)
    c = Case(E, O, "fem3d", (24000, 3, 20, 20, 13500, 1, 4), E.make_config(**kw))
    on = E.Plan(c.m, E.make_config(col_sharing=1, **kw), upload=False)
    off = E.Plan(c.m, E.make_config(col_sharing=2, **kw), upload=False)
    s_on, s_off = on.stats, off.stats
    # sharing off: one word per stored pair and (valid) lane
    assert 0.93 * s_off["size_block_ell"] <= s_off["col_words"] * 2 <= s_off["size_block_ell"]
    assert s_on["col_words"] < 0.36 * s_off["col_words"]              # groups of 3 (+ partial groups at slab edges)
    assert s_on["size_block_ell"] == s_off["size_block_ell"] and s_on["nnz_er"] == s_off["nnz_er"]
    g = (on.array("slab_meta").reshape(-1, 4)[:, 3] & 0x3F) + 1
    assert g.max() <= 26 and np.median(g) == 22                        # 64 lanes = 21 nodes x 3 rows + 1
    assert ((off.array("slab_meta").reshape(-1, 4)[:, 3] & 0x3F) + 1).min() >= 1
    y_on, _ = O.walk_plan(on, c.xp)
    y_off, _ = O.walk_plan(off, c.xp)
    assert np.array_equal(y_on, y_off)
    assert c.check(y_on)[0] == 0
    assert s_on["bytes_format"] < 0.90 * s_off["bytes_format"]


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_reference_window_rule_matches_restated_convert(E, O, seed):
    """EHYB_WINDOW_REFERENCE uses the membership test of convert.c:247: the residual count must
    equal toER of the restated COO2EHYB, and the ELL size must equal its sizeBlockELL at slab
    height 64 (widths rounded up to even here)."""
    rng = np.random.default_rng(seed)
    n, size, cache = 1024, 256, 256
    rows, cols = [], []
    for i in range(n):
        p = i // size
        for j in set(rng.integers(p * size, (p + 1) * size, rng.integers(1, 9)).tolist()
                     + rng.integers(0, n, rng.integers(0, 3)).tolist() + [i]):
            rows.append(i)
            cols.append(j)
    A = sp.coo_matrix((rng.uniform(-1, 1, len(rows)), (rows, cols)), shape=(n, n)).tocsr()
    A.sort_indices()
    # the reference has no working long-row rule; 2 doubles of the LDS budget hold the slab counter
    cfg = E.make_config(window_mode=1, lds_doubles=cache + 2, hub_rule=2)
    m = E.Matrix.from_csr(A.indptr, A.indices, A.data, cfg)
    pb = np.arange(0, n + 1, size, dtype=np.int32)
    m.c.nParts = len(pb) - 1
    m.part_boundary[:] = pb
    plan = E.Plan(m, cfg, upload=False)
    st = plan.stats
    L = R.build_reference_ehyb(A.indptr, A.indices, A.data, pb, cache, warp=64)
    assert st["nnz_er"] == L["to_er"]
    assert st["rows_er"] == L["rows_er"]
    # same widths up to the even rounding
    w_ref = L["width"]
    w_here = 2 * np.diff(plan.array("slab_pair_ptr").astype(np.int64))
    assert np.array_equal(w_here, (w_ref + 1) // 2 * 2)
    x = O.x_glibc(n)
    assert np.allclose(O.walk_plan(plan, x)[0], R.walk_reference_ehyb(L, x), rtol=0, atol=1e-13)


def _case_from_csr(E, O, A, cfg):
    A = sp.csr_matrix(A)
    A.sort_indices()
    m = E.Matrix.from_csr(A.indptr, A.indices, A.data, cfg)
    x = O.x_glibc(m.n)
    y_ref = O.spmv_coo(m.n, m.I, m.J, m.V, x) if m.nnz else np.zeros(m.n)
    scale = O.abs_rowsum(m.n, m.I, m.J, m.V, x) if m.nnz else np.zeros(m.n)
    return m, x, y_ref, scale


@pytest.mark.parametrize("mode", [1, 2, 3], ids=["refwindow", "halo", "halo+symmetric-pairs"])
def test_edge_cases(E, O, mode):
    """Empty rows, a dense row, a matrix smaller than one slab, a zero matrix, diagonal only -- and,
    for symmetric pair storage, symmetric matrices with dense rows/columns, repeated coordinates and
    a partner that differs in the last bit."""
    cfg = E.make_config(window_mode=min(mode, 2), lds_doubles=128 if mode < 3 else 256, er_seg_len=64, sym_pairs=int(mode == 3))
    rng = np.random.default_rng(0)
    S = sp.random(400, 400, density=0.02, random_state=7, format="csr")
    S = (S + S.T).tocsr()
    arrow = sp.lil_matrix((300, 300))
    arrow[0, :] = 1.5
    arrow[:, 0] = 1.5
    arrow.setdiag(4.0)
    almost = S.copy().tolil()
    i0, j0 = np.argwhere(S.toarray() != 0)[5]
    if i0 != j0:
        almost[i0, j0] = np.nextafter(S[i0, j0], 10.0)     # a_ij and a_ji differ in the last bit: must stay two entries
    mats = {
        "sym_random": S,
        "sym_arrow": arrow.tocsr(),
        "sym_last_bit": almost.tocsr(),
        "tiny_3x3": sp.csr_matrix(np.array([[1.0, 0, 2], [0, 0, 0], [3, 0, 4]])),
        "one_by_one": sp.csr_matrix(np.array([[2.5]])),
        "zero_5x5": sp.csr_matrix((5, 5)),
        "diag_200": sp.identity(200, format="csr") * 3.0,
        "empty_rows": sp.random(300, 300, density=0.01, random_state=1, format="csr"),
        "dense_row": sp.vstack([sp.csr_matrix(np.ones((1, 700))), sp.random(699, 700, density=0.005, random_state=2)]).tocsr(),
        "dense_col": sp.hstack([sp.csr_matrix(np.ones((500, 1))), sp.random(500, 499, density=0.005, random_state=3)]).tocsr(),
    }
    for name, A in mats.items():
        m, x, y_ref, scale = _case_from_csr(E, O, A, cfg)
        if m.nnz:
            m.reorder(cfg, symmetric=False)
        else:
            m.reorder_list[:] = np.arange(m.n, dtype=np.int32)
            m.c.nParts = 1
            m.part_boundary[:] = [0, m.n]
        plan = E.Plan(m, cfg, upload=False)
        yp, written = O.walk_plan(plan, E.vector_reorder(x, m.reorder_list))
        assert (written == 1).all(), name
        y = E.vector_recover(yp, m.reorder_list)
        assert O.check_tolerance(y, y_ref, scale + 1e-300)[0] == 0, name
    del rng


def test_symmetric_pairs_with_repeated_coordinates(E, O):
    """A coordinate that appears more than once (the reference's reader would keep both,
    solver_test.c:96-103): every copy is either paired with exactly one equal partner or kept as it is."""
    n = 130
    rows, cols, vals = [], [], []
    rng = np.random.default_rng(3)
    for i in range(n):
        rows.append(i), cols.append(i), vals.append(5.0 + i * 1e-3)
    for _ in range(400):
        i, j = rng.integers(0, n, 2)
        if i == j:
            continue
        v = float(rng.integers(1, 9)) / 8
        copies = int(rng.integers(1, 4))                      # (i,j) up to three times ...
        mirror = int(rng.integers(0, 4))                      # ... and (j,i) a different number of times
        rows += [i] * copies + [j] * mirror
        cols += [j] * copies + [i] * mirror
        vals += [v] * (copies + mirror)
    order = np.argsort(np.array(rows), kind="stable")
    r, c, v = np.array(rows)[order], np.array(cols)[order], np.array(vals)[order]
    indptr = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(np.bincount(r, minlength=n), out=indptr[1:])
    cfg = E.make_config(lds_doubles=256, sym_pairs=1)
    m = E.Matrix.from_csr(indptr, c, v, cfg)
    x = O.x_glibc(n)
    y_ref = O.spmv_coo(n, m.I, m.J, m.V, x)
    scale = O.abs_rowsum(n, m.I, m.J, m.V, x)
    m.reorder(cfg, symmetric=False)
    plan = E.Plan(m, cfg, upload=False)
    assert plan.stats["sym_pairs"] > 0
    yp, written = O.walk_plan(plan, E.vector_reorder(x, m.reorder_list))
    assert (written == 1).all()
    assert O.check_tolerance(E.vector_recover(yp, m.reorder_list), y_ref, scale)[0] == 0


def test_oversized_partition_is_split(E, O):
    """A caller whose partitions exceed the window (e.g. a single partition) still gets a valid plan."""
    cfg = E.make_config(window_mode=2, lds_doubles=256)
    c = Case(E, O, "stencil2d", (60, 50, 5, 500, 1), cfg, reorder=False)
    c.m.c.nParts = 1
    c.m.part_boundary[:] = [0, c.n]
    plan = E.Plan(c.m, cfg, upload=False)
    assert plan.stats["n_parts"] >= c.n // 256
    _walk_ok(E, O, c, plan)
    # no partition information at all
    c.m.c.nParts = 0
    plan2 = E.Plan(c.m, cfg, upload=False)
    _walk_ok(E, O, c, plan2)


def test_row_range_plans_compose(E, O):
    """Plans over row blocks (the multi-GPU sharding) together reproduce the whole product."""
    cfg = E.make_config(window_mode=2, lds_doubles=512, n_top=2)
    c = Case(E, O, "fem3d", (12000, 3, 16, 16, 13500, 1, 3), cfg)
    pb = c.m.part_boundary
    cut = int(pb[len(pb) // 2])
    y = np.zeros(c.n)
    for r0, r1 in ((0, cut), (cut, c.n)):
        plan = E.Plan(c.m, cfg, rows=(r0, r1), upload=False)
        yp, written = O.walk_plan(plan, c.xp)
        assert written[r0:r1].min() == 1 and written[:r0].sum() == 0 and written[r1:].sum() == 0
        y[r0:r1] = yp[r0:r1]
        # with n_top > 1 the ELL windows only touch the block's own x segment
        hc = plan.array("halo_cols")
        assert len(hc) == 0 or (hc.min() >= r0 and hc.max() < r1)
    assert c.check(y)[0] == 0
    with pytest.raises(E.EhybError):
        E.Plan(c.m, cfg, rows=(1, c.n), upload=False)  # not on a partition boundary


def test_bad_inputs_are_rejected(E, O):
    cfg = E.make_config()
    A = sp.random(50, 50, density=0.1, random_state=0, format="csr")
    m = E.Matrix.from_csr(A.indptr, A.indices, A.data, cfg)
    m.c.nParts = 0
    keep = m.J[0]
    m.J[0] = 50  # out of range column
    with pytest.raises(E.EhybError):
        E.Plan(m, cfg, upload=False)
    m.J[0] = keep
    m.row_idx[3] = m.row_idx[2] - 1 if m.row_idx[2] > 0 else 0
    m.row_idx[2] = m.row_idx[3] + 5
    with pytest.raises(E.EhybError):
        E.Plan(m, cfg, upload=False)


@pytest.mark.parametrize("case", ["fem3d", "stencil", "rmat", "pattern-symmetric-only"])
def test_symmetric_pair_storage(E, O, case):
    """cfg.sym_pairs: an in-partition pair a_ij == a_ji is stored once and the owning lane also adds
    a_ij * x_i to row j's accumulator.  The layout must stand for exactly the same matrix whatever the
    input looks like: symmetric (most in-partition entries pair up), unsymmetric or symmetric in
    pattern only (nothing pairs up, nothing breaks)."""
    cfg = E.make_config(lds_doubles=2048, sym_pairs=1)
    if case == "fem3d":
        c = Case(E, O, "fem3d", (9000, 3, 12, 12, 20000, 1, 5), cfg)
    elif case == "stencil":
        c = Case(E, O, "stencil2d", (70, 60, 9, 500, 3), cfg)
    elif case == "rmat":
        c = Case(E, O, "rmat", (12, 1 << 15, 2), cfg)
    else:
        c = Case(E, O, "stencil2d", (70, 60, 5, 0, 3), cfg)
        v = c.m.V
        v[:] = np.arange(1, len(v) + 1) * 1e-3          # a_ij != a_ji everywhere off the diagonal
        c = c.refresh_reference() if hasattr(c, "refresh_reference") else c
    plan = E.Plan(c.m, cfg, upload=False)
    st = plan.stats
    assert st["nnz_ell"] + st["nnz_er"] == st["nnz"] == c.m.nnz
    assert st["size_block_ell"] - st["ell_padding"] + st["sym_pairs"] == st["nnz_ell"]   # stored + mirrored = represented
    items = plan.array("items").reshape(-1, 8)
    assert len(items) == st["n_parts"] and np.all(items[:, 1] - items[:, 0] == 1)         # one workgroup per partition
    assert st["lds_bytes"] <= cfg.lds_doubles * 8
    if case in ("fem3d", "stencil"):
        assert st["sym_pairs"] > 0.25 * st["nnz"], "most in-partition entries of a symmetric matrix pair up"
        # balanced orientation: the rows of a slab stay about equally long
        assert st["ell_padding"] < (0.25 if case == "fem3d" else 0.4) * st["size_block_ell"]   # 9-point rows: 5 or 6 stored entries in 3 pairs
    if case == "pattern-symmetric-only":
        assert st["sym_pairs"] == 0
    if case != "pattern-symmetric-only":
        _walk_ok(E, O, c, plan)
    else:
        yp, written = O.walk_plan(plan, c.xp)
        A = c.m.to_scipy()
        assert np.allclose(yp, A @ c.xp, rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("kind,args,kw", [
    ("fem3d", (24000, 3, 20, 20, 13500, 1, 3), dict(sym_pairs=1, value_map=1, lds_doubles=4096)),
    ("fem3d", (24000, 3, 20, 20, 13500, 1, 3), dict(lds_doubles=2048)),
    ("fem3d_graded", (30000, 3, 20, 20, 100000, 705000, 1, 1), dict(sym_pairs=1, lds_doubles=4096)),
    ("rmat", (13, 1 << 16, 1), dict(lds_doubles=2048)),
], ids=["fem-sym", "fem-plain", "graded-sym", "rmat"])
def test_column_maps_and_sorted_lists_build_the_same_plan(E, O, kind, args, kw):
    """cfg.col_map: the host builder finds a window's outside columns through one look-up array per thread (default) or through
    sorted lists / hash tables and binary searches (col_map = 2, the only route for inputs whose arrays would not fit): the
    permutation and every array of the plan are the same."""
    from ehyb_spmv_gpu_amd.host import ARRAYS
    got = []
    for col_map in (1, 2):
        cfg = E.make_config(col_map=col_map, **kw)
        m = E.Matrix.generate(kind, *args, cfg=cfg)
        m.reorder(cfg)
        plan = E.Plan(m, cfg, upload=False)
        got.append((m.reorder_list.copy(), {name: plan.array(name) for name in ARRAYS}, plan.stats))
    assert np.array_equal(got[0][0], got[1][0])
    assert got[0][2] == got[1][2]
    for name in got[0][1]:
        assert np.array_equal(got[0][1][name], got[1][1][name]), name
