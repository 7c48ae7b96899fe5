"""The reference's own driver against this repo's headers and library: solver_test.c
compiles UNCHANGED against include/ (kernel.h, spmv.h, reordering.h), links against libehyb.so and
runs up to the device boundary, where -- on a box without a GPU -- the product refuses loudly.
Also: what the reference-named entry points do with the sizing hints of the reference's driver,
the partBoundary capacity contract, and the caller's OpenMP setting.  No GPU compute here."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
DRIVER = os.path.join(ROOT, "oracle", "_ref", "solver_test_ref")


def _driver():
    if os.path.isdir(REF):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "ref"], check=True)
    if not os.path.exists(DRIVER):
        pytest.skip("neither the reference tree nor a prebuilt oracle/_ref/solver_test_ref is present")
    return DRIVER


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present")
def test_reference_driver_compiles_unchanged_against_include():
    """g++ on /root/reference/solver_test.c as it lies (fed on stdin so that "kernel.h" resolves to
    include/kernel.h), + the reference's mmio.c, -lehyb: the undefined symbols it leaves are exactly
    the boundary of SURVEY 8b."""
    exe = _driver()
    und = subprocess.run(["nm", "-u", "-C", exe], capture_output=True, text=True, check=True).stdout
    for sym in ("matrixReorder(_matrixCOO*)", "matrixReorder_unsym(_matrixCOO*)", "spmvGPuEHYB",
                "vectorReorder(int, double const*, double*, int const*)",
                "vectorRecover(int, double const*, double*, int const*)"):
        assert sym in und, sym
    assert "cuda" not in und.lower() and "cusparse" not in und.lower() and "MTMETIS" not in und
    # kernel.h carries the constants the driver's sizing uses (reference kernel.h:20-28), as C and as C++
    for std, cc in (("-std=c99", "gcc"), ("-std=c++17", "g++")):
        p = subprocess.run([cc, std, "-x", "c" if cc == "gcc" else "c++", "-fsyntax-only", "-Wall", "-Werror", "-Wno-unused",
                            "-I", os.path.join(ROOT, "include"), "-"], text=True, capture_output=True,
                           input='#include "kernel.h"\nint a[smSize == 82 && smSize2 == 80 && threadELL == 1024 && (maxSharedMem) == 95232 ? 1 : -1];\n')
        assert p.returncode == 0, p.stderr


def test_reference_driver_runs_to_the_device_boundary(E, tmp_path):
    """./spmv.out -m <name> -i <iters> (reference README.md:10) with the reference's main(): read,
    its own sizing, its own CPU product, then this library's reorder step -- and, without a GPU, a loud
    refusal at spmvGPuEHYB (status 4 = EHYB_ERR_NO_DEVICE).  With a GPU the whole chain is
    tests/test_gpu_dropin.py."""
    exe = _driver()
    if E.device_count() > 0:
        pytest.skip("a GPU is visible: covered by tests/test_gpu_dropin.py")
    (tmp_path / "read").mkdir()
    E.Matrix.generate("fem3d", 36000, 3, 24, 24, 13500, 1, 1).write_mtx(tmp_path / "read" / "s.mtx", symmetric_lower_only=True)
    E.Matrix.generate("rmat", 15, 1 << 17, 3).write_mtx(tmp_path / "read" / "g.mtx")
    for name, banner in (("s", "read symmetric matrix"), ("g", "unsymmetric reordering")):
        p = subprocess.run([exe, "-m", name, "-i", "5"], cwd=tmp_path, capture_output=True, text=True, timeout=300)
        assert banner in p.stdout and "parts is 10 with cachSize 8192" in p.stdout   # solver_test.c:78,183: its own sizing
        assert p.returncode == 4 and "no CPU fallback" in p.stderr, (p.returncode, p.stderr[-500:])


def test_matrix_reorder_resizes_the_reference_hints(E):
    """matrixReorder(m) with the reference driver's sizing for audikw_1-class input (nParts = 164,
    cache 6144: solver_test.c:158-182) re-derives the partitions for 256 CUs and symmetric pair
    storage and writes what it used back; spmvGPuEHYB's own storage choice recognises them."""
    from ehyb_spmv_gpu_amd import _lib

    lib = _lib.load()
    m = E.Matrix.generate("fem3d", 120000, 3, 35, 35, 13500, 1, 1)
    m.c.nParts, m.c.vectorCacheSize, m.c.kernelPerPart = 164, 6144, 0
    x = E.x_glibc(m.n)
    ref = m.to_scipy() @ x
    m.reorder_dropin()
    assert m.c.nParts >= 128 and m.c.nParts != 164
    pb = m.part_boundary
    assert pb[0] == 0 and pb[-1] == m.n and np.all(np.diff(pb) >= 0)
    cfg_sym = E.make_config(sym_pairs=1)
    assert 2 * (int(np.diff(pb).max()) + 1) <= cfg_sym.lds_doubles - 2
    # same matrix, permuted
    perm = m.reorder_list
    assert sorted(perm.tolist()) == list(range(m.n))
    y = E.vector_recover(m.to_scipy() @ E.vector_reorder(x, perm), perm)
    assert np.allclose(y, ref, rtol=0, atol=1e-12 * np.abs(ref).max() + 1e-15)
    plan = E.Plan(m, cfg_sym, upload=False)        # what spmvGPuEHYB builds for such a matrix
    assert plan.stats["sym_pairs"] > 0.3 * plan.stats["nnz"] / 2
    # an unsymmetric matrix through matrixReorder_unsym keeps plain-storage partitions
    g = E.Matrix.generate("rmat", 16, 1 << 19, 3)
    g.c.nParts, g.c.vectorCacheSize = 82, 1024
    g.reorder_dropin()
    assert 2 * (int(np.diff(g.part_boundary).max()) + 1) > cfg_sym.lds_doubles - 2 or g.c.nParts < 128
    del lib


def test_matrix_reorder_allocates_part_boundary_like_the_reference(E):
    """matrixReorder / matrixReorder_unsym malloc m->partBoundary themselves (reordering.c:44,234): a caller that
    sized its own array for ITS nParts (the driver's 10 partitions) or passes NULL must not see a write through its
    pointer, whatever partition count comes back."""
    for gen, sym in ((("fem3d", 60000, 3, 28, 28, 13500, 1, 1), True), (("rmat", 15, 1 << 18, 3), False)):
        m = E.Matrix.generate(*gen)
        m.symmetric = sym
        lib_owned = m._lib_part_boundary = C.cast(m.c.partBoundary, C.c_void_p).value
        small = (C.c_int * 16)(*([-7] * 16))          # 10 partitions + 1, plus guard words
        m.c.nParts, m.c.vectorCacheSize, m.c.kernelPerPart = 10, 8192, 8
        m.c.partBoundary = C.cast(small, C.POINTER(C.c_int))
        m.reorder_dropin()
        assert list(small) == [-7] * 16
        assert C.cast(m.c.partBoundary, C.c_void_p).value != C.addressof(small)
        pb = m.part_boundary
        assert m.c.nParts >= 1 and pb[0] == 0 and pb[-1] == m.n and np.all(np.diff(pb) >= 0)
        C.CDLL(None).free(C.c_void_p(lib_owned))      # the generator's array, replaced above by `small`
        # NULL is legal too
        m2 = E.Matrix.generate(*gen)
        m2.symmetric = sym
        C.CDLL(None).free(C.cast(m2.c.partBoundary, C.c_void_p))
        m2.c.partBoundary = C.cast(None, C.POINTER(C.c_int))
        m2.reorder_dropin()
        assert m2.part_boundary[-1] == m2.n


def test_part_boundary_capacity_is_respected(E):
    """A C caller that holds exactly nParts+1 boundaries (the reference contract, spmv.h:31) and does
    not say otherwise never gets more partitions back; with a stated capacity the capacity split may
    add some, never beyond it."""
    from ehyb_spmv_gpu_amd import _lib

    lib = _lib.load()
    cfg = E.make_config(lds_doubles=2048, part_rows=1024)
    m = E.Matrix.generate("fem3d", 30000, 3, 22, 22, 13500, 1, 1, cfg=cfg)
    asked = int(m.c.nParts)
    # guard words behind the nParts+1 entries the caller "owns"
    pbv = np.frombuffer((C.c_char * (4 * (m.n + 1))).from_address(C.addressof(m.c.partBoundary.contents)), dtype=np.int32)
    pbv[asked + 1:] = -7
    c0 = E.make_config(lds_doubles=2048, part_rows=1024)          # part_boundary_cap = 0: unknown
    assert lib.ehyb_matrix_reorder(C.byref(m.c), 1, C.byref(c0)) == 0
    assert m.c.nParts == asked and np.all(pbv[asked + 1:] == -7)
    m2 = E.Matrix.generate("fem3d", 30000, 3, 22, 22, 13500, 1, 1, cfg=cfg)
    m2.reorder(cfg)                                              # host.py states dimension + 1
    assert m2.c.nParts >= asked
    # two-level partition without room: refused, not overflowed
    m3 = E.Matrix.generate("fem3d", 30000, 3, 22, 22, 13500, 1, 1, cfg=cfg)
    m3.c.nParts = 2
    c3 = E.make_config(lds_doubles=2048, part_rows=1024, n_top=2)
    assert lib.ehyb_matrix_reorder(C.byref(m3.c), 1, C.byref(c3)) == 1   # EHYB_ERR_ARG
    assert b"partBoundary" in lib.ehyb_last_error()


def test_callers_openmp_setting_survives(E):
    """The library's parallel regions use cfg.host_threads; the caller's own OpenMP thread count is
    what it was when the call returns (no load-time or per-call global change)."""
    omp = C.CDLL("libgomp.so.1")
    omp.omp_get_max_threads.restype = C.c_int
    omp.omp_set_num_threads(3)
    cfg = E.make_config(host_threads=2)
    m = E.Matrix.generate("fem3d", 30000, 3, 22, 22, 13500, 1, 1, cfg=cfg)
    assert omp.omp_get_max_threads() == 3
    m.reorder(cfg)
    assert omp.omp_get_max_threads() == 3
    E.Plan(m, cfg, upload=False)
    assert omp.omp_get_max_threads() == 3
    omp.omp_set_num_threads(E.host_threads())
