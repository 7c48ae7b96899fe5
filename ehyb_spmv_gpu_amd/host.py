"""Python mirror of the reference's host interface for the EHYB path.

Names follow the reference (solver_test.c / spmv.h / reordering.h): a `Matrix` is a
`matrixCOO`, `matrix_reorder` is `matrixReorder[_unsym]`, `vector_reorder` /
`vector_recover` are `vectorReorder` / `vectorRecover`, `spmv_gpu_ehyb` is `spmvGPuEHYB`.
Everything is a thin call into libehyb.so; numpy is used only to hand buffers over.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import Config, MatrixCOO, Stats

EHYB_WINDOW_REFERENCE = 1
EHYB_WINDOW_HALO = 2
EHYB_PART_AUTO, EHYB_PART_CONTIGUOUS, EHYB_PART_MULTILEVEL, EHYB_PART_MTMETIS, EHYB_PART_DEGREE = 0, 1, 2, 3, 4

ARRAYS = {
    "part_boundary": (0, np.int32), "win_len": (1, np.int32), "halo_ptr": (2, np.int32),
    "halo_cols": (3, np.int32), "slab_pair_ptr": (4, np.uint32), "slab_row": (5, np.int32),
    "slab_part": (6, np.int32), "ell_val": (7, np.float64), "ell_col": (8, np.uint32),
    "items": (9, np.int32), "er_seg_ptr": (10, np.int64), "er_seg_row": (11, np.int32),
    "er_col": (12, np.int32), "er_val": (13, np.float64), "er_bins": (14, np.int32),
    "slab_col_ptr": (15, np.uint32), "lane_group": (16, np.uint8), "slab_meta": (17, np.uint32),
    "segs": (18, np.int32), "perm": (19, np.int32), "slab_lrow": (20, np.uint16),
    "pb_val": (21, np.float64), "pb_col": (22, np.uint16), "pb_dst": (23, np.uint32), "pb_units1": (24, np.int32),
    "pb_row": (25, np.uint16), "pb_units2": (26, np.int32),
    "pb_colf": (31, np.uint16), "pb_chunk": (32, np.uint32), "pb_jump": (33, np.uint32),
    "ell_src": (27, np.int32), "er_src": (28, np.int32), "pb_src": (29, np.int32), "ell_src2": (30, np.int32),
    "col_seg_first": (34, np.int32), "pb_seg_item": (35, np.int32), "pb_items1": (36, np.int32),
}


class EhybError(RuntimeError):
    def __init__(self, code, where):
        msg = _lib.load().ehyb_last_error().decode(errors="replace")
        super().__init__(f"{where} failed with status {code}: {msg}")
        self.code = code


def _check(code, where):
    if code != 0:
        raise EhybError(code, where)


def make_config(**kw):
    """ehyb_config with defaults, overridden by keyword (field names of ehyb.h)."""
    cfg = Config()  # all zero = all defaults
    for k, v in kw.items():
        if not hasattr(cfg, k):
            raise TypeError(f"unknown ehyb_config field {k!r}")
        setattr(cfg, k, int(v))
    # fill in what the library will use (window and partition sizes depend on the mode fields)
    _lib.load().ehyb_config_resolve(C.byref(cfg), C.byref(cfg))
    return cfg


def _ptr(a, ctype):
    return a.ctypes.data_as(C.POINTER(ctype))


def _view(ptr, count, dtype):
    if count == 0:
        return np.zeros(0, dtype=dtype)
    buf = (C.c_char * (count * np.dtype(dtype).itemsize)).from_address(C.addressof(ptr.contents))
    return np.frombuffer(buf, dtype=dtype, count=count)


def x_glibc(n):
    """x[i]: srand(i); (rand()%200-100)/1000 -- solver_test.c:89-92, 228-231."""
    x = np.empty(n, dtype=np.float64)
    _lib.load().ehyb_x_glibc(n, _ptr(x, C.c_double))
    return x


def sizing(dimension, cfg=None):
    """(nParts, vectorCacheSize, kernelPerPart) -- solver_test.c:53-77 re-derived."""
    a, b, c = C.c_int(), C.c_int(), C.c_int()
    _check(_lib.load().ehyb_sizing(dimension, C.byref(cfg) if cfg else None, C.byref(a), C.byref(b), C.byref(c)),
           "ehyb_sizing")
    return a.value, b.value, c.value


class Matrix:
    """Owner of a matrixCOO whose arrays live in the C heap (the reorder step frees and
    replaces I/J/V exactly like reordering.c:363-369)."""

    def __init__(self):
        self.lib = _lib.load()
        self.c = MatrixCOO()
        self.symmetric = False
        self._alive = False

    # ---- constructors
    @classmethod
    def from_csr(cls, indptr, indices, data, cfg=None, symmetric=False):
        m = cls()
        indptr = np.ascontiguousarray(indptr, dtype=np.int64)
        indices = np.ascontiguousarray(indices, dtype=np.int32)
        data = np.ascontiguousarray(data, dtype=np.float64)
        n = len(indptr) - 1
        _check(m.lib.ehyb_matrix_from_csr(n, _ptr(indptr, C.c_int64), _ptr(indices, C.c_int), _ptr(data, C.c_double),
                                          C.byref(cfg) if cfg else None, C.byref(m.c)), "ehyb_matrix_from_csr")
        m._alive = True
        m.symmetric = symmetric
        return m

    @classmethod
    def read_mtx(cls, path, cfg=None):
        m = cls()
        sym = C.c_int(0)
        _check(m.lib.ehyb_mm_read(str(path).encode(), C.byref(cfg) if cfg else None, C.byref(m.c), C.byref(sym)),
               "ehyb_mm_read")
        m._alive = True
        m.symmetric = bool(sym.value)
        return m

    @classmethod
    def generate(cls, kind, *args, cfg=None):
        m = cls()
        cp = C.byref(cfg) if cfg else None
        if kind == "banded":
            n, band, block = args
            rc = m.lib.ehyb_gen_banded(n, band, block, cp, C.byref(m.c))
        elif kind == "fem3d":
            n, dof, nx, ny, ppm, scramble, seed = args
            rc = m.lib.ehyb_gen_fem3d(n, dof, nx, ny, ppm, scramble, seed, cp, C.byref(m.c))
            m.symmetric = True
        elif kind == "fem3d_graded":
            n, dof, nx, ny, near_ppm, far_ppm, scramble, seed = args
            rc = m.lib.ehyb_gen_fem3d_graded(n, dof, nx, ny, near_ppm, far_ppm, scramble, seed, cp, C.byref(m.c))
            m.symmetric = True
        elif kind == "fem3d_block":
            n, dof, nx, ny, ppm, scramble, seed, block, n_blocks = args
            rc = m.lib.ehyb_gen_fem3d_block(n, dof, nx, ny, ppm, scramble, seed, block, n_blocks, cp, C.byref(m.c))
            m.symmetric = n_blocks == 1
        elif kind == "rmat":
            scale, edges, seed = args
            rc = m.lib.ehyb_gen_rmat(scale, edges, seed, cp, C.byref(m.c))
        elif kind == "rmat_block":
            scale, edges, seed, block, n_blocks = args[:5]
            cost_model = int(args[5]) if len(args) > 5 else 0      # 1: cuts balanced for the "cover" exchange (ehyb_gen_rmat_block_cost)
            cuts = (C.c_int * (n_blocks + 1))()
            rc = m.lib.ehyb_gen_rmat_block_cost(scale, edges, seed, block, n_blocks, cost_model, cuts, cp, C.byref(m.c))
            m.block_cuts = [int(c) for c in cuts]
        elif kind == "rmat_rows":
            scale, edges, seed, row0, row1 = args
            rc = m.lib.ehyb_gen_rmat_rows(scale, edges, seed, row0, row1, cp, C.byref(m.c))
        elif kind == "stencil2d":
            nx, ny, points, extra, seed = args
            rc = m.lib.ehyb_gen_stencil2d(nx, ny, points, extra, seed, cp, C.byref(m.c))
            m.symmetric = True
        elif kind == "kkt3d":
            (nx,) = args
            rc = m.lib.ehyb_gen_kkt3d(nx, cp, C.byref(m.c))
            m.symmetric = True
        elif kind == "mesh3d":
            n, dof, knn, grade, seed = args
            rc = m.lib.ehyb_gen_mesh3d(n, dof, knn, grade, seed, cp, C.byref(m.c))
            m.symmetric = True
        else:
            raise ValueError(f"unknown generator {kind!r}")
        _check(rc, f"ehyb_gen_{kind}")
        m._alive = True
        return m

    # ---- views (valid until reorder()/free())
    @property
    def n(self):
        return self.c.dimension

    @property
    def nnz(self):
        return self.c.totalNum

    def _arr(self, name, count, dtype):
        return _view(getattr(self.c, name), count, dtype)

    @property
    def I(self):
        return self._arr("I", self.nnz, np.int32)

    @property
    def J(self):
        return self._arr("J", self.nnz, np.int32)

    @property
    def V(self):
        return self._arr("V", self.nnz, np.float64)

    @property
    def row_idx(self):
        return self._arr("rowIdx", self.n + 1, np.int32)

    @property
    def num_in_row(self):
        return self._arr("numInRow", self.n, np.int32)

    @property
    def num_in_row2(self):
        return self._arr("numInRow2", self.n, np.int32)

    @property
    def part_boundary(self):
        return self._arr("partBoundary", self.c.nParts + 1, np.int32)

    @property
    def reorder_list(self):
        return self._arr("reorderList", self.n, np.int32)

    def to_scipy(self):
        import scipy.sparse as sp

        return sp.csr_matrix((self.V.copy(), self.J.copy(), self.row_idx.astype(np.int64)), shape=(self.n, self.n))

    # ---- the pre-step (solver_test.c:369-376)
    def reorder(self, cfg=None, symmetric=None):
        """matrixReorder / matrixReorder_unsym (reordering.c:231-378 / 41-228), in place.
        With cfg.n_top > 1 the first partition of every top-level block is kept in self.block_first."""
        sym = self.symmetric if symmetric is None else symmetric
        c = Config.from_buffer_copy(cfg) if cfg is not None else Config()
        c.part_boundary_cap = self.n + 1  # what ehyb_mm_read / ehyb_gen_* / ehyb_matrix_from_csr allocate
        n_top = max(1, int(c.n_top))
        blocks = (C.c_int * (n_top + 1))()
        _check(self.lib.ehyb_matrix_reorder_blocks(C.byref(self.c), 1 if sym else 0, C.byref(c), blocks),
               "ehyb_matrix_reorder")
        self.block_first = [int(b) for b in blocks]
        return self

    def reorder_dropin(self):
        """matrixReorder(m) / matrixReorder_unsym(m) exactly as the reference's driver calls them
        (solver_test.c:369-374; the C++-linkage symbols of include/reordering.h): no configuration,
        nParts / vectorCacheSize as the caller left them are only hints."""
        name = "_Z13matrixReorderP10_matrixCOO" if self.symmetric else "_Z19matrixReorder_unsymP10_matrixCOO"
        old = C.cast(self.c.partBoundary, C.c_void_p).value
        if not hasattr(self, "_lib_part_boundary"):
            self._lib_part_boundary = old          # first call: what the reader / generator allocated
        getattr(self.lib, name)(C.byref(self.c))
        # like the reference (reordering.c:44,234) the call mallocs a new partBoundary over the field; the array this
        # library's reader/generator had put there is ours to release
        if old and old == getattr(self, "_lib_part_boundary", None) and old != C.cast(self.c.partBoundary, C.c_void_p).value:
            C.CDLL(None).free(C.c_void_p(old))
        return self

    def key(self):
        """ehyb_matrix_key: digest of the matrix as stored (take it BEFORE reorder() to key a plan cache)."""
        return int(self.lib.ehyb_matrix_key(C.byref(self.c)))

    def append_ghosts(self, n_ghost, gi, gj, gv):
        """Rank-local multi-GPU build: add n_ghost receive-buffer columns and the entries coupling
        to them (ehyb_matrix_append_ghosts)."""
        gi = np.ascontiguousarray(gi, dtype=np.int32)
        gj = np.ascontiguousarray(gj, dtype=np.int32)
        gv = np.ascontiguousarray(gv, dtype=np.float64)
        _check(self.lib.ehyb_matrix_append_ghosts(C.byref(self.c), int(n_ghost), len(gi), _ptr(gi, C.c_int), _ptr(gj, C.c_int),
                                                  _ptr(gv, C.c_double)), "ehyb_matrix_append_ghosts")
        self.symmetric = False
        return self

    def write_mtx(self, path, symmetric_lower_only=False):
        _check(self.lib.ehyb_mm_write(str(path).encode(), C.byref(self.c), int(symmetric_lower_only)), "ehyb_mm_write")

    def free(self):
        if self._alive:
            self.lib.ehyb_matrix_free(C.byref(self.c))
            self._alive = False

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def vector_reorder(v, reorder_list):
    """vectorReorder (reordering.c:380-384): out[list[i]] = v[i]."""
    v = np.ascontiguousarray(v, dtype=np.float64)
    lst = np.ascontiguousarray(reorder_list, dtype=np.int32)
    out = np.empty_like(v)
    _lib.load().ehyb_vector_reorder(len(v), _ptr(v, C.c_double), _ptr(out, C.c_double), _ptr(lst, C.c_int))
    return out


def entry_order(row_idx_before, reorder_list):
    """ehyb_entry_order: out[k_new] = k_old, the place of every entry of the reordered matrix in the caller's
    arrays before ehyb_matrix_reorder (the V scatter of reordering.c:348-362 as an index list)."""
    rp = np.ascontiguousarray(row_idx_before, dtype=np.int32)
    lst = np.ascontiguousarray(reorder_list, dtype=np.int32)
    n = len(lst)
    assert rp.shape == (n + 1,)
    out = np.empty(int(rp[-1]) - int(rp[0]), dtype=np.int32)
    _check(_lib.load().ehyb_entry_order(n, _ptr(rp, C.c_int), _ptr(lst, C.c_int), _ptr(out, C.c_int32)), "ehyb_entry_order")
    return out


def vector_recover(v_rodr, reorder_list):
    """vectorRecover (reordering.c:386-391): out[i] = v_rodr[list[i]]."""
    v = np.ascontiguousarray(v_rodr, dtype=np.float64)
    lst = np.ascontiguousarray(reorder_list, dtype=np.int32)
    out = np.empty_like(v)
    _lib.load().ehyb_vector_recover(len(v), _ptr(v, C.c_double), _ptr(out, C.c_double), _ptr(lst, C.c_int))
    return out


def partition_graph(indptr, indices, nparts, max_part_rows=0, vwgt=None, cfg=None):
    """ehyb_partition_graph: stands in for MTMETIS_PartGraphKway (reordering.c:280-293)."""
    indptr = np.ascontiguousarray(indptr, dtype=np.int64)
    indices = np.ascontiguousarray(indices, dtype=np.int32)
    n = len(indptr) - 1
    part = np.empty(n, dtype=np.int32)
    cut = C.c_int64(0)
    vw = None if vwgt is None else _ptr(np.ascontiguousarray(vwgt, dtype=np.int32), C.c_int)
    _check(_lib.load().ehyb_partition_graph(n, _ptr(indptr, C.c_int64), _ptr(indices, C.c_int), vw, nparts,
                                            max_part_rows, C.byref(cfg) if cfg else None, _ptr(part, C.c_int),
                                            C.byref(cut)), "ehyb_partition_graph")
    return part, cut.value


class Plan:
    """ehyb_plan: the device-resident EHYB layout of one (permuted) matrix."""

    def __init__(self, matrix, cfg=None, rows=None, upload=True, col_segs=None):
        """col_segs (multi-GPU): first column of every column segment + the dimension (ehyb_plan_create_host_segs);
        the multiply can then run segment by segment (spmv_part)."""
        self.lib = _lib.load()
        self.h = C.c_void_p()
        self.n = matrix.n
        r0, r1 = (0, matrix.n) if rows is None else rows
        self.rows = (r0, r1)
        segs = None if col_segs is None else np.ascontiguousarray(col_segs, dtype=np.int32)
        args = (C.byref(matrix.c), r0, r1, C.byref(cfg) if cfg else None, 0 if segs is None else len(segs) - 1,
                None if segs is None else _ptr(segs, C.c_int), C.byref(self.h))
        if upload:
            # build + upload in one call: what the device can build (cfg.symbolic: the panel form of a residual) is built there
            _check(self.lib.ehyb_plan_create_segs(*args), "ehyb_plan_create_segs")
        else:
            _check(self.lib.ehyb_plan_create_host_segs(*args), "ehyb_plan_create_host_segs")

    def upload(self):
        _check(self.lib.ehyb_plan_upload(self.h), "ehyb_plan_upload")
        return self

    def save(self, path, reorder_list=None, key=0):
        """ehyb_plan_save: the host layout (+ the permutation, + a matrix key) to a cache file."""
        lst = None if reorder_list is None else np.ascontiguousarray(reorder_list, dtype=np.int32)
        assert lst is None or lst.shape == (self.n,)
        _check(self.lib.ehyb_plan_save(self.h, _ptr(lst, C.c_int) if lst is not None else None, key, str(path).encode()),
               "ehyb_plan_save")

    @classmethod
    def load(cls, path, key=0, want_perm=True, upload=True):
        """ehyb_plan_load -> (plan, reorder_list or None); key != 0 must match the file's."""
        self = cls.__new__(cls)
        self.lib = _lib.load()
        self.h = C.c_void_p()
        _check(self.lib.ehyb_plan_load(str(path).encode(), key, C.byref(self.h), None), "ehyb_plan_load")
        st = self.stats
        self.n = st["n_cols"]
        seg = self.array("part_boundary")
        self.rows = (int(seg[0]), int(seg[-1]))
        perm = self.array("perm") if want_perm else None
        if perm is not None and len(perm) == 0:
            perm = None
        if upload:
            self.upload()
        return self, perm

    @property
    def stats(self):
        st = Stats()
        _check(self.lib.ehyb_plan_stats(self.h, C.byref(st)), "ehyb_plan_stats")
        return st.as_dict()

    def array(self, name):
        which, dtype = ARRAYS[name]
        p = C.c_void_p()
        cnt = C.c_int64()
        _check(self.lib.ehyb_plan_host_array(self.h, which, C.byref(p), C.byref(cnt)), "ehyb_plan_host_array")
        if cnt.value == 0 or not p.value:
            return np.zeros(0, dtype=dtype)
        buf = (C.c_char * (cnt.value * np.dtype(dtype).itemsize)).from_address(p.value)
        return np.frombuffer(buf, dtype=dtype, count=cnt.value).copy()

    def spmv(self, x_dev, y_dev, stream=0, phase=0, walk=None):
        """Asynchronous y = A x on device pointers (ints).  walk: None = the plan's own alternation (ehyb_spmv), 0 / 1 = this
        launch walks its streams first to last / last to first (ehyb_spmv_walk)."""
        if walk is not None:
            _check(self.lib.ehyb_spmv_walk(self.h, C.c_void_p(x_dev), C.c_void_p(y_dev), C.c_void_p(stream), int(walk)), "ehyb_spmv_walk")
            return
        _check(self.lib.ehyb_spmv_phase(self.h, C.c_void_p(x_dev), C.c_void_p(y_dev), C.c_void_p(stream), phase),
               "ehyb_spmv")

    def graph(self, x_dev, y_dev, multiplies=1):
        """ehyb_spmv_graph_create: `multiplies` multiplies captured into a hipGraph that keeps the alternating walk -> SpmvGraph"""
        return SpmvGraph(self, x_dev, y_dev, multiplies)

    def tune(self, x_dev, y_dev, reps=5):
        """ehyb_plan_tune: the heaviest work items on the XCDs measured fastest -> (launch span before, after) in us."""
        a, b = C.c_double(0), C.c_double(0)
        _check(self.lib.ehyb_plan_tune(self.h, C.c_void_p(x_dev), C.c_void_p(y_dev), reps, C.byref(a), C.byref(b)), "ehyb_plan_tune")
        return a.value, b.value

    def spmv_part(self, x_dev, y_dev, stream, seg_begin, seg_end, flags):
        """ehyb_spmv_part: flags 1 = the ELL launch first, 2 = the closing pass (EHYB_PART_FIRST / _LAST)."""
        _check(self.lib.ehyb_spmv_part(self.h, C.c_void_p(x_dev), C.c_void_p(y_dev), C.c_void_p(stream), seg_begin, seg_end, flags),
               "ehyb_spmv_part")

    @property
    def col_segs(self):
        n = C.c_int(0)
        _check(self.lib.ehyb_plan_col_segs(self.h, C.byref(n)), "ehyb_plan_col_segs")
        return n.value

    def spmv_host(self, x, iters=1):
        x = np.ascontiguousarray(x, dtype=np.float64)
        assert x.shape == (self.n,)
        y = np.zeros(self.n, dtype=np.float64)
        _check(self.lib.ehyb_spmv_host(self.h, _ptr(x, C.c_double), _ptr(y, C.c_double), iters), "ehyb_spmv_host")
        return y

    def set_values(self, values, entry_order=None, stream=0):
        """ehyb_plan_set_values: the numeric phase of the build again, on the device, for new values on the
        plan's pattern (needs cfg.value_map = 1).  values / entry_order: numpy arrays (host) or, both of them,
        device pointers given as (ptr, count) tuples."""
        if isinstance(values, tuple):
            ptr, count = values
            order = None if entry_order is None else C.c_void_p(entry_order[0])
            _check(self.lib.ehyb_plan_set_values(self.h, C.c_void_p(ptr), count, order, 1, C.c_void_p(stream)), "ehyb_plan_set_values")
            return
        v = np.ascontiguousarray(values, dtype=np.float64)
        o = None if entry_order is None else np.ascontiguousarray(entry_order, dtype=np.int32)
        assert o is None or o.shape == v.shape
        _check(self.lib.ehyb_plan_set_values(self.h, v.ctypes.data_as(C.c_void_p), len(v), o.ctypes.data_as(C.c_void_p) if o is not None else None,
                                             0, C.c_void_p(stream)), "ehyb_plan_set_values")

    def bench(self, x_dev, y_dev, stream=0, warmup=10, iters=100, per_kernel=True):
        t, e, r = C.c_double(), C.c_double(), C.c_double()
        _check(self.lib.ehyb_spmv_bench(self.h, C.c_void_p(x_dev), C.c_void_p(y_dev), C.c_void_p(stream), warmup, iters,
                                        C.byref(t), C.byref(e) if per_kernel else None,
                                        C.byref(r) if per_kernel else None), "ehyb_spmv_bench")
        return {"ms_total": t.value, "ms_ell_avg": e.value, "ms_er_avg": r.value}

    def cg(self, b, x0=None, max_iter=1000, rtol=1e-10, check_every=10, inv_diag=None):
        """ehyb_cg / ehyb_pcg: (Jacobi-preconditioned if inv_diag is given) conjugate gradients on the
        device; b, x, inv_diag in the permuted numbering.  -> (x, iterations, relative residual)"""
        b = np.ascontiguousarray(b, dtype=np.float64)
        db, dx = DeviceBuffer(self.n).upload(b), DeviceBuffer(self.n)
        dx.upload(np.zeros(self.n) if x0 is None else x0)
        dd = None if inv_diag is None else DeviceBuffer(self.n).upload(np.ascontiguousarray(inv_diag, dtype=np.float64))
        it, rel = C.c_int(0), C.c_double(0)
        _check(self.lib.ehyb_pcg(self.h, C.c_void_p(dd.ptr) if dd else None, C.c_void_p(db.ptr), C.c_void_p(dx.ptr),
                                 max_iter, rtol, check_every, None, C.byref(it), C.byref(rel)), "ehyb_pcg")
        return dx.download(), it.value, rel.value

    def destroy(self):
        if self.h:
            self.lib.ehyb_plan_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


def spmv_gpu_ehyb(matrix, vector_in, max_iter, cfg=None, timing=False):
    """spmvGPuEHYB (spmv.cu:61-133): y = A_perm * x on the GPU, 10 warm-ups + max_iter runs.
    cfg None = what the drop-in symbol does by itself (storage chosen from the matrix).
    timing=True also returns the milliseconds of the max_iter timed multiplies."""
    lib = _lib.load()
    x = np.ascontiguousarray(vector_in, dtype=np.float64)
    y = np.zeros(matrix.n, dtype=np.float64)
    it = C.c_int(0)
    ms = C.c_double(0)
    _check(lib.spmvGPuEHYB_cfg(C.byref(matrix.c), _ptr(x, C.c_double), _ptr(y, C.c_double), max_iter, C.byref(it),
                               C.byref(cfg) if cfg is not None else None, C.byref(ms)), "spmvGPuEHYB")
    return (y, it.value, ms.value) if timing else (y, it.value)


def host_threads():
    """ehyb_host_threads: OpenMP threads of the host builder = the CPUs the process owns (affinity, cgroup quota)."""
    return int(_lib.load().ehyb_host_threads())


def device_count():
    c = C.c_int(0)
    _lib.load().ehyb_device_count(C.byref(c))
    return c.value


class SpmvGraph:
    """A captured run of multiplies of one plan (include/ehyb.h: ehyb_spmv_graph_create / ehyb_graph_launch)."""

    def __init__(self, plan, x_dev, y_dev, multiplies=1):
        self.lib = plan.lib
        self.h = C.c_void_p()
        _check(self.lib.ehyb_spmv_graph_create(plan.h, C.c_void_p(x_dev), C.c_void_p(y_dev), int(multiplies), C.byref(self.h)), "ehyb_spmv_graph_create")

    def launch(self, stream=0):
        _check(self.lib.ehyb_graph_launch(self.h, C.c_void_p(stream)), "ehyb_graph_launch")

    def destroy(self):
        if self.h:
            self.lib.ehyb_graph_destroy(self.h)
            self.h = None


class Stream:
    """ehyb_stream_create: a non-blocking HIP stream for callers without a HIP binding (tests)."""

    def __init__(self):
        self.lib = _lib.load()
        self.p = C.c_void_p()
        _check(self.lib.ehyb_stream_create(C.byref(self.p)), "ehyb_stream_create")

    @property
    def ptr(self):
        return self.p.value

    def sync(self):
        _check(self.lib.ehyb_stream_sync(self.p), "ehyb_stream_sync")

    def destroy(self):
        if self.p:
            self.lib.ehyb_stream_destroy(self.p)
            self.p = None


class DeviceBuffer:
    """hipMalloc'ed array of doubles (tests / CLI plumbing; bench.py uses torch tensors)."""

    def __init__(self, n):
        self.lib = _lib.load()
        self.n = n
        self.p = C.c_void_p()
        _check(self.lib.ehyb_dev_alloc(n * 8, C.byref(self.p)), "ehyb_dev_alloc")

    @property
    def ptr(self):
        return self.p.value

    def upload(self, a):
        a = np.ascontiguousarray(a, dtype=np.float64)
        _check(self.lib.ehyb_h2d(self.p, a.ctypes.data_as(C.c_void_p), a.nbytes), "ehyb_h2d")
        return self

    def download(self):
        out = np.empty(self.n, dtype=np.float64)
        _check(self.lib.ehyb_d2h(out.ctypes.data_as(C.c_void_p), self.p, out.nbytes), "ehyb_d2h")
        return out

    def free(self):
        if self.p:
            self.lib.ehyb_dev_free(self.p)
            self.p = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass
