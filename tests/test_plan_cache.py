"""On-disk plan cache (ehyb_plan_save / ehyb_plan_load / ehyb_matrix_key): the
permutation and the finished layout survive a round trip bit for bit, and files that are damaged,
from another matrix or not plan files at all are refused before anything reaches a GPU."""
import numpy as np
import pytest

from util import Case


def _plan(E, O, tmp_path, **kw):
    cfg = E.make_config(lds_doubles=1024, **kw)
    raw = E.Matrix.generate("fem3d", 9000, 3, 12, 12, 20000, 1, 11, cfg=cfg)
    key = raw.key()
    c = Case(E, O, "fem3d", (9000, 3, 12, 12, 20000, 1, 11), cfg)       # generated again and reordered
    plan = E.Plan(c.m, cfg, upload=False)
    path = tmp_path / "plan.ehyb"
    plan.save(path, c.m.reorder_list, key)
    return c, plan, path, key


@pytest.mark.parametrize("fuse", [0, 2], ids=["inline-residual", "csr-residual"])
def test_round_trip_is_bit_exact(E, O, tmp_path, fuse):
    c, plan, path, key = _plan(E, O, tmp_path, fuse_er=fuse, cap_split=2)
    loaded, perm = E.Plan.load(path, key=key, upload=False)
    assert np.array_equal(perm, c.m.reorder_list)
    assert loaded.stats == plan.stats and loaded.n == plan.n and loaded.rows == plan.rows
    for name in E.ARRAYS:
        if name == "perm":
            continue
        a, b = plan.array(name), loaded.array(name)
        assert a.dtype == b.dtype and np.array_equal(a, b), name
    # the loaded layout multiplies like the one it was saved from
    y0, _ = O.walk_plan(plan, c.xp)
    y1, written = O.walk_plan(loaded, c.xp)
    assert np.array_equal(y0, y1) and written.min() == 1
    assert c.check(y1)[0] == 0


def test_key_depends_on_the_matrix(E, O):
    a = E.Matrix.generate("stencil2d", 40, 30, 5, 100, 3)
    b = E.Matrix.generate("stencil2d", 40, 30, 5, 100, 3)
    c = E.Matrix.generate("stencil2d", 40, 30, 5, 100, 4)
    assert a.key() == b.key() != 0
    assert a.key() != c.key()
    v = b.V
    v[7] = np.nextafter(v[7], 1.0)          # one ulp in one value
    assert a.key() != b.key()


def test_bad_files_are_refused(E, O, tmp_path):
    c, plan, path, key = _plan(E, O, tmp_path)
    with pytest.raises(E.EhybError) as e:
        E.Plan.load(path, key=key ^ 1, upload=False)                       # another matrix
    assert e.value.code == 6 and "another matrix" in str(e.value)
    loaded, _ = E.Plan.load(path, key=0, upload=False)                     # 0 = do not check
    assert loaded.stats == plan.stats
    data = path.read_bytes()
    cut = tmp_path / "cut.ehyb"
    cut.write_bytes(data[:len(data) // 2])
    with pytest.raises(E.EhybError) as e:
        E.Plan.load(cut, upload=False)
    assert e.value.code == 6
    junk = tmp_path / "junk.ehyb"
    junk.write_bytes(b"%%MatrixMarket matrix coordinate real general\n" * 10)
    with pytest.raises(E.EhybError) as e:
        E.Plan.load(junk, upload=False)
    assert e.value.code == 6
    # a flipped count in the middle of the file: sizes no longer fit together
    bad = bytearray(data)
    perm_bytes = np.ascontiguousarray(c.perm, dtype=np.int32).tobytes()
    off = data.index(perm_bytes) - 8                      # the 64-bit element count in front of the stored permutation
    assert int.from_bytes(data[off:off + 8], "little") == c.n
    bad[off:off + 8] = (int.from_bytes(bad[off:off + 8], "little") - 3).to_bytes(8, "little")
    (tmp_path / "bad.ehyb").write_bytes(bytes(bad))
    with pytest.raises(E.EhybError) as e:
        E.Plan.load(tmp_path / "bad.ehyb", upload=False)
    assert e.value.code == 6
    with pytest.raises(E.EhybError) as e:
        E.Plan.load(tmp_path / "missing.ehyb", upload=False)
    assert e.value.code == 5


def test_saved_without_permutation(E, O, tmp_path):
    c, plan, path, key = _plan(E, O, tmp_path)
    plan.save(tmp_path / "noperm.ehyb")
    loaded, perm = E.Plan.load(tmp_path / "noperm.ehyb", upload=False)
    assert perm is None and loaded.stats == plan.stats


@pytest.mark.gpu
def test_loaded_plan_multiplies_identically_on_the_gpu(E, O, gpu, tmp_path):
    c, plan, path, key = _plan(E, O, tmp_path)
    plan.upload()
    loaded, perm = E.Plan.load(path, key=key)
    y0, y1 = plan.spmv_host(c.xp), loaded.spmv_host(c.xp)
    assert np.array_equal(y0, y1) and c.check(y1)[0] == 0
