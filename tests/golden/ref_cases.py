"""The matrices behind tests/golden/ref_driver_<tag>.txt: built by the deterministic generators (plus deterministic edits),
written to ./read/a.mtx for the reference's unchanged driver by make_ref_driver_golden.py and rebuilt by
tests/test_golden.py.  Every case has more than 30,010 rows: the driver prints rows 30000..30009 (solver_test.c:385-388).

  sym            FEM, 3 unknowns per node, symmetric file (lower triangle)       -- matrixRead_sym, solver_test.c:127-265
  general        R-MAT, general file                                             -- matrixRead_unsym, solver_test.c:31-126
  general_zeros  banded general file with EXPLICIT ZEROS (every 7th stored value) and values of mixed magnitude: the
                 general reader keeps file order and counts explicit zeros as entries (solver_test.c:96-103)
  sym_zeros      9-point stencil + random couplings, symmetric file whose lower triangle holds explicit zeros off the
                 diagonal AND zero diagonal entries (every 13th row; the symmetric reader needs every diagonal entry
                 present: totalNum = 2*stored - dimension, solver_test.c:150-156) -- the mirrored expansion of
                 solver_test.c:235-255 with entries that contribute nothing
"""
import numpy as np


def build(E, tag):
    """-> (Matrix, symmetric_file)"""
    if tag == "sym":
        return E.Matrix.generate("fem3d", 120000, 3, 35, 35, 13500, 1, 1), True
    if tag == "general":
        return E.Matrix.generate("rmat", 16, 1 << 19, 3), False
    if tag == "general_zeros":
        m = E.Matrix.generate("banded", 40960, 24, 1024)
        V, I, J = m.V, m.I, m.J
        k = np.arange(len(V))
        V[k % 7 == 3] = 0.0                                   # explicit zeros, kept as entries
        big = (I.astype(np.int64) * 31 + J) % 5 == 0
        V[big] *= 1000.0                                      # mixed magnitudes: sums that nearly cancel
        return m, False
    if tag == "sym_zeros":
        m = E.Matrix.generate("stencil2d", 210, 200, 9, 5000, 5)
        V, I, J = m.V, m.I.astype(np.int64), m.J.astype(np.int64)
        lo, hi = np.minimum(I, J), np.maximum(I, J)
        V[(lo != hi) & ((lo * 7 + hi * 3) % 11 == 0)] = 0.0   # symmetric: a pair goes together
        V[(lo == hi) & (lo % 13 == 0)] = 0.0                  # stored zero diagonal entries
        return m, True
    raise KeyError(tag)


TAGS = ("sym", "general", "general_zeros", "sym_zeros")
