"""The oracle against everything that can pin it (CPU only).

The reference has no golden vectors and cannot be built here (oracle/ehyb_oracle.c header), so
the pins are: glibc known answers for the x rule, scipy.sparse as an independent product, the
committed fixtures of tests/golden, and self-consistency of the restated reference format
(build + kernel walk == COO product).
"""
import json
import os

import numpy as np
import pytest
import scipy.sparse as sp

from oracle import ehyb_ref_layout as R

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_x_rule_known_answers(O, E):
    """solver_test.c:89-92 under glibc: values recorded in SURVEY.md section 4."""
    known = json.load(open(os.path.join(GOLD, "x_known.json")))
    x = O.x_glibc(943695)
    for k, v in known.items():
        if k.isdigit():
            assert x[int(k)] == pytest.approx(v, abs=1e-15), k
    assert x[:10974].sum() == pytest.approx(known["sum_0_10973"], abs=1e-9)
    assert np.all(x >= -0.1) and np.all(x < 0.1)
    # the product's harness helper follows the same rule
    assert np.array_equal(E.x_glibc(5000), x[:5000])


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_products_agree_with_scipy(O, seed):
    rng = np.random.default_rng(seed)
    n = 700
    A = sp.random(n, n, density=0.02, random_state=seed, format="coo", dtype=np.float64)
    order = rng.permutation(A.nnz)  # storage order must not matter beyond rounding
    I, J, V = A.row[order].astype(np.int32), A.col[order].astype(np.int32), A.data[order]
    x = O.x_glibc(n)
    y = O.spmv_coo(n, I, J, V, x)
    scale = O.abs_rowsum(n, I, J, V, x)
    assert O.check_tolerance(y, A.tocsr() @ x, scale)[0] == 0
    C = A.tocsr()
    for omp in (False, True):
        yc = O.spmv_csr(C.indptr, C.indices, C.data, x, omp=omp)
        assert O.check_tolerance(yc, y, scale)[0] == 0
    assert np.allclose(scale, abs(C) @ abs(x), rtol=1e-13)


def test_symmetric_rule_equals_expanded(O):
    """solver_test.c:235-255: applying each stored lower entry twice == product with the expanded matrix."""
    rng = np.random.default_rng(5)
    n = 300
    L = sp.tril(sp.random(n, n, density=0.05, random_state=3), format="coo")
    x = O.x_glibc(n)
    y = O.spmv_sym_lower(n, L.row.astype(np.int32), L.col.astype(np.int32), L.data, x)
    full = L + sp.tril(L, -1).T
    scale = abs(full) @ abs(x)
    assert O.check_tolerance(y, full @ x, scale)[0] == 0
    del rng


def test_compare_restated(O):
    """compare() of solver_test.c:7-29."""
    a = np.array([1.0, 2.0, 0.0, -4.0])
    b = np.array([1.0, 2.05, 0.0, -4.0])
    bad, diff, ampl = O.compare(a, b, 0.01)
    assert bad == 1
    assert diff == pytest.approx(0.05)
    assert ampl == pytest.approx(0.05 / 2.0)
    assert O.compare(a, a, 0.01) == (0, 0.0, 0.0)


def test_reference_sizing_known_cases():
    """solver_test.c:158-182 restated: the two cases SURVEY.md 8a/8d quotes."""
    assert R.reference_sizing(10974, True) == (10, 8192, 8)       # bcsstk17
    assert R.reference_sizing(943695, True)[:2] == (164, 6144)    # audikw_1
    # int16 overflow of vectorCacheSize past ~2.6 M rows
    assert R.reference_sizing(4194304, True)[1] < 0 or R.reference_sizing(4194304, True)[1] * 8 <= 93 * 1024


def _small_matrix(seed, n=640, nparts=4, cache=192):
    rng = np.random.default_rng(seed)
    rows, cols = [], []
    size = n // nparts
    for i in range(n):
        p = i // size
        k = rng.integers(1, 12)
        local = rng.integers(p * size, (p + 1) * size, k)
        far = rng.integers(0, n, rng.integers(0, 3))
        for j in set(local.tolist() + far.tolist() + [i]):
            rows.append(i)
            cols.append(j)
    A = sp.coo_matrix((rng.uniform(-1, 1, len(rows)), (rows, cols)), shape=(n, n)).tocsr()
    A.sort_indices()
    pb = np.arange(0, n + 1, size)
    return A, pb, cache


@pytest.mark.parametrize("warp", [32, 64])
@pytest.mark.parametrize("seed", [0, 1])
def test_reference_format_walk_equals_product(O, warp, seed):
    """Restated COO2EHYB + kernel walk (convert.c / kernel.cu) reproduces the CPU product."""
    A, pb, cache = _small_matrix(seed)
    n = A.shape[0]
    x = O.x_glibc(n)
    L = R.build_reference_ehyb(A.indptr, A.indices, A.data, pb, cache, warp=warp)
    y = R.walk_reference_ehyb(L, x)
    I = np.repeat(np.arange(n), np.diff(A.indptr)).astype(np.int32)
    y_ref = O.spmv_coo(n, I, A.indices, A.data, x)
    scale = O.abs_rowsum(n, I, A.indices, A.data, x)
    assert O.check_tolerance(y, y_ref, scale)[0] == 0
    assert L["nnz_ell"] + L["to_er"] == A.nnz
    assert L["size_block_ell"] == L["nnz_ell"] + L["waste"]
    assert L["to_er"] > 0  # the generator leaves some entries outside the window
