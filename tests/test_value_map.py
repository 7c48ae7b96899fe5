"""Slot maps of the value streams (cfg.value_map; csrc/layout.cpp, er_panel.cpp) and ehyb_entry_order: the host
half of the device-side numeric phase (ehyb_plan_set_values, csrc/ehyb_fill.hip -- SURVEY 8f-2; the reference
fills valBlockELL / valER on the host, convert.c:316-369, after the V scatter of reordering.c:348-362).
Checked without a GPU: gathering the matrix's own values through the maps reproduces every value stream bit
for bit, every entry is covered, and a slot of symmetric pair storage names both entries it stands for."""
import numpy as np
import pytest

from util import Case

CASES = [
    ("fem_halo", "fem3d", (30000, 3, 22, 22, 13500, 1, 1), dict(lds_doubles=4096, direct=2)),
    ("fem_sym", "fem3d", (30000, 3, 22, 22, 13500, 1, 1), dict(lds_doubles=4096, sym_pairs=1)),
    ("fem_inline_residual", "fem3d", (30000, 3, 22, 22, 13500, 1, 1), dict(lds_doubles=2048, fuse_er=1, direct=2)),
    ("fem_reference_window", "fem3d", (30000, 3, 22, 22, 13500, 1, 1), dict(window_mode=1, lds_doubles=1024, fuse_er=2)),
    ("rmat_csr_residual", "rmat", (14, 1 << 17, 1), dict(lds_doubles=512, er_mode=1, direct=2)),
    ("rmat_panel_residual", "rmat", (14, 1 << 17, 1), dict(lds_doubles=512, er_mode=2, fuse_er=2, er_panel_cols=512, er_block_rows=300)),
    ("rmat_pruned_windows", "rmat", (16, 1 << 19, 3), dict(er_mode=2, direct=2)),
    ("small_direct", "fem3d", (12000, 3, 16, 16, 13500, 1, 3), dict(direct=1)),
    ("banded_relative_columns", "banded", (1 << 14, 32, 1024), dict(lds_doubles=2048, partitioner=1, direct=2)),
    ("kkt_sym", "kkt3d", (14,), dict(lds_doubles=2048, sym_pairs=1)),
]


def gather(V, src):
    out = np.zeros(len(src))
    ok = src >= 0
    out[ok] = V[src[ok]]
    return out


@pytest.mark.parametrize("name,kind,args,kw", CASES, ids=[c[0] for c in CASES])
def test_maps_reproduce_the_value_streams(E, O, name, kind, args, kw):
    cfg = E.make_config(value_map=1, **kw)
    c = Case(E, O, kind, args, cfg)
    V = c.m.V.copy()
    plan = E.Plan(c.m, cfg, upload=False)
    st = plan.stats
    ell_val, ell_src = plan.array("ell_val"), plan.array("ell_src")
    er_val, er_src = plan.array("er_val"), plan.array("er_src")
    pb_val, pb_src = plan.array("pb_val"), plan.array("pb_src")
    assert ell_src.shape == ell_val.shape and er_src.shape == er_val.shape and pb_src.shape == pb_val.shape
    assert np.array_equal(gather(V, ell_src).view(np.int64), ell_val.view(np.int64))
    assert np.array_equal(gather(V, er_src).view(np.int64), er_val.view(np.int64))
    assert np.array_equal(gather(V, pb_src).view(np.int64), pb_val.view(np.int64))
    assert er_src.size == 0 or er_src.min() >= 0
    # every entry of the matrix is in exactly one place: an ELL slot, the mirror of one, or the residual
    # (the inline form keeps its residual entries in both the ELL stream and the CSR segments)
    seen = np.zeros(c.nnz, dtype=np.int64)
    np.add.at(seen, ell_src[ell_src >= 0], 1)
    src2 = plan.array("ell_src2")
    if st["sym_pairs"] > 0:
        assert src2.shape == ell_val.shape
        both = src2 >= 0
        assert int(both.sum()) == st["sym_pairs"] and np.all(ell_src[both] >= 0)
        assert np.array_equal(V[src2[both]].view(np.int64), V[ell_src[both]].view(np.int64))   # a_ij == a_ji bitwise
        # the mirror entry really is the transposed one
        I, J = c.m.I, c.m.J
        assert np.array_equal(I[src2[both]], J[ell_src[both]]) and np.array_equal(J[src2[both]], I[ell_src[both]])
        np.add.at(seen, src2[both], 1)
    else:
        assert src2.size == 0
    if st["er_inline"] > 0:
        assert np.all(seen[er_src] == 1)
    else:
        np.add.at(seen, er_src, 1)
    assert seen.min() == 1 and seen.max() == 1
    if st["er_partials"] > 0:   # the panel form holds the same entries as the CSR segments
        assert np.array_equal(np.sort(pb_src[pb_src >= 0]), np.sort(er_src))


def test_no_maps_unless_asked(E, O):
    cfg = E.make_config(lds_doubles=4096, direct=2)
    c = Case(E, O, "fem3d", (30000, 3, 22, 22, 13500, 1, 1), cfg)
    plan = E.Plan(c.m, cfg, upload=False)
    for name in ("ell_src", "er_src", "pb_src", "ell_src2"):
        assert plan.array(name).size == 0


@pytest.mark.parametrize("kind,args,kw", [("fem3d", (30000, 3, 22, 22, 13500, 1, 1), dict(lds_doubles=4096)),
                                          ("rmat", (14, 1 << 17, 1), dict(lds_doubles=512)),
                                          ("fem3d", (30000, 3, 22, 22, 13500, 1, 1), dict(lds_doubles=4096, n_top=3))],
                         ids=["fem", "rmat", "two_level"])
def test_entry_order_is_the_scatter_of_the_reorder_step(E, O, kind, args, kw):
    """V after ehyb_matrix_reorder == V before it, gathered through ehyb_entry_order (reordering.c:348-362)."""
    cfg = E.make_config(**kw)
    m = E.Matrix.generate(kind, *args, cfg=cfg)
    I0, J0, V0, rp0 = m.I.copy(), m.J.copy(), m.V.copy(), m.row_idx.copy()
    m.reorder(cfg)
    order = E.entry_order(rp0, m.reorder_list)
    assert np.array_equal(np.sort(order), np.arange(len(V0)))
    assert np.array_equal(m.V.view(np.int64), V0[order].view(np.int64))
    lst = m.reorder_list
    assert np.array_equal(m.I, lst[I0[order]]) and np.array_equal(m.J, lst[J0[order]])


def test_entry_order_refuses_a_non_permutation(E):
    with pytest.raises(E.EhybError):
        E.entry_order(np.array([0, 1, 2, 3], dtype=np.int32), np.array([0, 0, 1], dtype=np.int32))


@pytest.mark.parametrize("seed", range(500, 560))
def test_maps_on_random_plans(E, O, seed):
    """The slot maps under the random matrices and plan configurations of the layout fuzz (tests/fuzz_cases.py): every
    value stream is the matrix's values gathered through its map, whatever shape the plan took."""
    from fuzz_cases import build

    m, cfg, kw, x, y_ref, scale = build(E, O, seed)
    cfg.value_map = 1
    plan = E.Plan(m, cfg, upload=False)
    V = m.V.copy() if m.nnz else np.zeros(0)
    for val, src in (("ell_val", "ell_src"), ("er_val", "er_src"), ("pb_val", "pb_src")):
        v, s = plan.array(val), plan.array(src)
        assert v.shape == s.shape, (val, kw)
        assert np.array_equal(gather(V, s).view(np.int64), v.view(np.int64)), (val, kw)
    st = plan.stats
    s2 = plan.array("ell_src2")
    assert int((s2 >= 0).sum()) == st["sym_pairs"]
    if st["sym_pairs"]:
        es = plan.array("ell_src")
        both = s2 >= 0
        assert np.array_equal(V[s2[both]].view(np.int64), V[es[both]].view(np.int64))
