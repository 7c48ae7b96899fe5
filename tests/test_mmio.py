"""Matrix Market reader of the harness (ehyb_mm_read: the role of matrixRead_sym/_unsym,
solver_test.c:31-265) against scipy.io and against the reference's own mmio.c, which is
compiled from where it lies into oracle/_ref/libmmio_ref.so (test infrastructure only)."""
import ctypes as C
import os

import numpy as np
import pytest
import scipy.io
import scipy.sparse as sp

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")
REF_MMIO = os.path.join(os.path.dirname(HERE), "oracle", "_ref", "libmmio_ref.so")


def ref_banner_and_size(path):
    """mm_read_banner + mm_read_mtx_crd_size of the reference (mmio.c:96-217)."""
    lib = C.CDLL(REF_MMIO)
    libc = C.CDLL("libc.so.6")
    libc.fopen.restype = C.c_void_p
    libc.fopen.argtypes = [C.c_char_p, C.c_char_p]
    libc.fclose.argtypes = [C.c_void_p]
    lib.mm_read_banner.argtypes = [C.c_void_p, C.c_char_p]
    lib.mm_read_mtx_crd_size.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    f = libc.fopen(path.encode(), b"r")
    assert f
    code = C.create_string_buffer(4)
    rc = lib.mm_read_banner(f, code)
    m, n, nz = C.c_int(), C.c_int(), C.c_int()
    rc2 = lib.mm_read_mtx_crd_size(f, C.byref(m), C.byref(n), C.byref(nz)) if rc == 0 else -1
    libc.fclose(f)
    return rc, code.raw.decode(), rc2, m.value, n.value, nz.value


@pytest.mark.parametrize("name", ["sym_grid", "gen_band", "pat_graph"])
def test_reader_matches_scipy_and_reference_mmio(E, name):
    path = os.path.join(GOLD, name + ".mtx")
    m = E.Matrix.read_mtx(path)
    S = sp.csr_matrix(scipy.io.mmread(path))
    assert (m.n, m.nnz) == (S.shape[0], S.nnz)
    assert abs(m.to_scipy() - S).max() == 0
    assert np.array_equal(np.repeat(np.arange(m.n), np.diff(m.row_idx)), m.I)
    assert m.c.maxCol == np.diff(S.indptr).max()
    if not os.path.exists(REF_MMIO):
        pytest.skip("oracle/_ref/libmmio_ref.so not built (needs /root/reference at build time)")
    rc, code, rc2, M, N, nz = ref_banner_and_size(path)
    assert rc == 0 and rc2 == 0 and M == N == m.n
    assert (code[3] == "S") == m.symmetric                      # mm_is_symmetric, solver_test.c:348
    stored = sum(1 for line in open(path) if not line.startswith("%")) - 1
    assert nz == stored
    if m.symmetric:
        # solver_test.c:135: totalNum = 2*stored - dimension (assumes a full diagonal, as here)
        assert m.nnz == 2 * nz - m.n


def test_symmetric_placement_order(E, tmp_path):
    """Entries are placed in file order, the mirrored one right after its original
    (solver_test.c:235-255): row contents follow that order, not column order."""
    p = tmp_path / "s.mtx"
    p.write_text("%%MatrixMarket matrix coordinate real symmetric\n3 3 5\n1 1 1.0\n3 1 2.0\n2 2 3.0\n3 2 4.0\n3 3 5.0\n")
    m = E.Matrix.read_mtx(p)
    assert m.nnz == 7 and m.symmetric
    assert m.row_idx.tolist() == [0, 2, 4, 7]
    assert m.J.tolist() == [0, 2, 1, 2, 0, 1, 2]
    assert m.V.tolist() == [1.0, 2.0, 3.0, 4.0, 2.0, 4.0, 5.0]


def test_general_keeps_per_row_file_order(E, tmp_path):
    p = tmp_path / "g.mtx"
    p.write_text("%%MatrixMarket matrix coordinate real general\n% c\n3 3 4\n2 3 1.5\n1 1 2.5\n2 1 3.5\n3 3 -1\n")
    m = E.Matrix.read_mtx(p)
    assert not m.symmetric and m.nnz == 4
    assert m.I.tolist() == [0, 1, 1, 2] and m.J.tolist() == [0, 2, 0, 2]
    assert m.V.tolist() == [2.5, 1.5, 3.5, -1.0]


def test_skew_integer_and_roundtrip(E, tmp_path):
    p = tmp_path / "k.mtx"
    p.write_text("%%MatrixMarket matrix coordinate integer skew-symmetric\n3 3 2\n2 1 4\n3 2 -7\n")
    m = E.Matrix.read_mtx(p)
    assert abs(m.to_scipy() - sp.csr_matrix(scipy.io.mmread(str(p)))).max() == 0
    q = tmp_path / "out.mtx"
    g = E.Matrix.generate("stencil2d", 6, 5, 9, 10, 1)
    g.write_mtx(q, symmetric_lower_only=True)
    back = E.Matrix.read_mtx(q)
    assert back.symmetric and abs(back.to_scipy() - g.to_scipy()).max() == 0


@pytest.mark.parametrize("text,why", [
    ("%%MatrixMarket matrix coordinate complex general\n2 2 1\n1 1 1.0 0.0\n", "complex is rejected (solver_test.c:339-345)"),
    ("%%MatrixMarket matrix array real general\n2 2\n1\n2\n3\n4\n", "dense arrays are not coordinate files"),
    ("%%NotMatrixMarket matrix coordinate real general\n2 2 0\n", "bad banner"),
    ("%%MatrixMarket matrix coordinate real general\n2 3 1\n1 1 1.0\n", "rectangular"),
    ("%%MatrixMarket matrix coordinate real general\n2 2 2\n1 1 1.0\n", "truncated body"),
    ("%%MatrixMarket matrix coordinate real general\n2 2 1\n3 1 1.0\n", "index out of range"),
])
def test_reader_rejects(E, tmp_path, text, why):
    p = tmp_path / "bad.mtx"
    p.write_text(text)
    with pytest.raises(E.EhybError) as ei:
        E.Matrix.read_mtx(p)
    assert ei.value.code in (5, 6), why


def test_missing_file(E, tmp_path):
    with pytest.raises(E.EhybError) as ei:
        E.Matrix.read_mtx(tmp_path / "nope.mtx")
    assert ei.value.code == 5  # EHYB_ERR_IO; the reference prints "file read error" and exits


def test_generators_are_deterministic_and_shaped(E):
    a = E.Matrix.generate("fem3d", 3000, 3, 10, 10, 13500, 1, 7)
    b = E.Matrix.generate("fem3d", 3000, 3, 10, 10, 13500, 1, 7)
    assert np.array_equal(a.J, b.J) and np.array_equal(a.V, b.V)
    A = a.to_scipy()
    assert abs(A - A.T).max() == 0, "fem3d is symmetric"
    assert np.all(np.diff(a.row_idx) % 3 == 0), "3 unknowns per node: row lengths are multiples of 3"
    band = E.Matrix.generate("banded", 4096, 32, 1024)
    B = band.to_scipy()
    assert np.all(np.diff(band.row_idx) == 32)
    blk = np.arange(4096) // 1024
    C_ = B.tocoo()
    assert np.all(blk[C_.row] == blk[C_.col]), "config 3: every entry stays inside its 1024-row block"
    i, j = 5, 1024 * 0 + (5 - 16) % 1024
    assert B[i, j] == pytest.approx((((31 * i + 17 * j) % 200) - 100) / 1000 or 0.001)
    r = E.Matrix.generate("rmat", 10, 1 << 13, 1)
    assert r.n == 1024 and r.nnz <= 1 << 13 and np.all(np.diff(r.J[r.row_idx[0]:r.row_idx[1]]) > 0)
    k = E.Matrix.generate("kkt3d", 6)
    K = k.to_scipy()
    assert abs(K - K.T).max() == 0 and k.n == 2 * 6 ** 3
    assert k.nnz - K.count_nonzero() == 6 ** 3, "explicit zeros on the diagonal of the (2,2) block are stored entries"


def test_gzip_input_and_parallel_parse(E, tmp_path):
    """The reader takes gzip files (also by finding <name>.mtx.gz for <name>.mtx) and parses the body in
    line-aligned pieces in parallel: same matrix as the plain file, comment and blank lines in the
    body tolerated, a bad line reported by its entry number."""
    import gzip

    m = E.Matrix.generate("fem3d", 30000, 3, 22, 22, 13500, 1, 3)      # ~2.3 M entries: several pieces
    plain = tmp_path / "a.mtx"
    m.write_mtx(plain, symmetric_lower_only=True)
    text = plain.read_bytes()
    lines = text.split(b"\n")
    lines.insert(5000, b"% a comment in the body")
    lines.insert(9000, b"   ")
    text2 = b"\n".join(lines)
    (tmp_path / "b.mtx.gz").write_bytes(gzip.compress(text2, compresslevel=1))
    a = E.Matrix.read_mtx(plain)
    b = E.Matrix.read_mtx(tmp_path / "b.mtx.gz")
    c = E.Matrix.read_mtx(tmp_path / "b.mtx")                           # found as b.mtx.gz
    for other in (b, c):
        assert other.n == a.n and other.nnz == a.nnz and other.symmetric
        assert np.array_equal(a.I, other.I) and np.array_equal(a.J, other.J) and np.array_equal(a.V, other.V)
    assert a.nnz == m.nnz and np.array_equal(np.sort(a.J + a.I.astype(np.int64) * a.n), np.sort(m.J + m.I.astype(np.int64) * m.n))
    # entry 777777 damaged
    data = [ln for ln in text.split(b"\n")]
    data[2 + 777776] = b"12 x 3.0"
    (tmp_path / "bad.mtx").write_bytes(b"\n".join(data))
    with pytest.raises(E.EhybError) as e:
        E.Matrix.read_mtx(tmp_path / "bad.mtx")
    assert e.value.code == 6 and "bad entry 777777 " in str(e.value)
    # fewer entries than the size line promises
    (tmp_path / "short.mtx").write_bytes(b"\n".join(text.split(b"\n")[:1000]) + b"\n")
    with pytest.raises(E.EhybError) as e:
        E.Matrix.read_mtx(tmp_path / "short.mtx")
    assert e.value.code == 6 and "bad entry 999 " in str(e.value)


def test_number_parsing_is_exactly_strtod(E, tmp_path):
    """The reader parses the body itself (glibc's strtod costs 300-500 ns per 17-digit value, audikw_1 has 39 M lines): the fast
    path -- up to 19 digits and |exponent| <= 27 in one x87 extended operation, anything else handed to strtod -- must return what
    fscanf("%lg") returns (solver_test.c:97,197) for every spelling: checked bit for bit against libc's strtod."""
    import ctypes
    import random

    random.seed(11)
    vals = []
    for _ in range(60000):
        nd = random.randint(1, 21)
        digs = "".join(random.choice("0123456789") for _ in range(nd))
        pos = random.randint(0, nd)
        s = digs[:pos] + "." + digs[pos:] if random.random() < 0.8 else digs
        s = "0." if s == "." else s
        if random.random() < 0.6:
            s += random.choice("eE") + random.choice(["", "+", "-"]) + str(random.randint(0, 40 if random.random() < 0.9 else 320))
        vals.append(("-" if random.random() < 0.3 else "") + s)
    vals += ["1e23", "8.5e-324", "4.9e-324", "1.7976931348623157e308", "0.1", "123456789012345678", "9007199254740993", "9007199254740992.5",
             "1.0000000000000002220446049250313080847263336181640625", "2.2250738585072011e-308", "1e-400", "1e400", "5e-324", ".5", "5.",
             "+3.25", "inf", "-inf", "nan", "0x1p3", "1e", "1e+", "2.5e-", "7.e2", "00012.500e+001", "-0", "-0.0e5"]
    n = len(vals)
    path = tmp_path / "numbers.mtx"
    with open(path, "w") as f:
        f.write("%%MatrixMarket matrix coordinate real general\n")
        f.write(f"{n} {n} {n}\n")
        for i, s in enumerate(vals):
            f.write(f"{i + 1} {(i * 7) % n + 1} {s}\n")
    m = E.Matrix.read_mtx(path)
    libc = ctypes.CDLL(None)
    libc.strtod.restype, libc.strtod.argtypes = ctypes.c_double, [ctypes.c_char_p, ctypes.c_void_p]
    want = np.array([libc.strtod(s.encode(), None) for s in vals])
    got = m.V
    assert np.array_equal(np.isnan(got), np.isnan(want))
    ok = ~np.isnan(want)
    assert np.array_equal(got[ok].view(np.int64), want[ok].view(np.int64))     # bit for bit, the sign of zero included


def test_comment_and_blank_lines_inside_the_body(E, tmp_path):
    """The fast line count takes every line of the body for an entry; a body with comments or blank lines is noticed while parsing
    and counted exactly."""
    path = tmp_path / "gaps.mtx"
    rows = 30000
    with open(path, "w") as f:
        f.write("%%MatrixMarket matrix coordinate real symmetric\n% a comment\n")
        f.write(f"{rows} {rows} {2 * rows - 1}\n")
        for i in range(rows):
            f.write(f"{i + 1} {i + 1} {2.0 + i}\n")
            if i % 1000 == 500:
                f.write("% comment inside the body\n\n   \n")
            if i:
                f.write(f"{i + 1} {i} -1.5\n")
    m = E.Matrix.read_mtx(path)
    assert m.symmetric and m.n == rows and m.nnz == 3 * rows - 2
    assert np.array_equal(m.I[:3], [0, 0, 1]) and np.array_equal(m.J[:4], [0, 1, 1, 0])     # row 0: (0,0), the mirror of (1,0); row 1: (1,1), (1,0)
    d = np.zeros(rows)
    np.add.at(d, m.I[m.I == m.J], m.V[m.I == m.J])
    assert np.array_equal(d, 2.0 + np.arange(rows))
    assert np.all(m.V[m.I != m.J] == -1.5) and (m.I != m.J).sum() == 2 * (rows - 1)
