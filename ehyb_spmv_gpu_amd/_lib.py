"""ctypes declarations for libehyb.so (include/ehyb.h, include/spmv.h).

The library is the product; this module only loads it and fails loudly when it is
missing.  There is no Python or CPU implementation of the multiply to fall back to.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# EHYB_LIB: another build of the same library (A/B timing of kernel changes on one GPU box)
LIB_PATH = os.environ.get("EHYB_LIB") or os.path.join(_HERE, "libehyb.so")


class MatrixCOO(C.Structure):
    """matrixCOO of include/spmv.h (layout of reference spmv.h:17-33)."""

    _fields_ = [
        ("totalNum", C.c_int),
        ("dimension", C.c_int),
        ("maxCol", C.c_int),
        ("nParts", C.c_int),
        ("vectorCacheSize", C.c_uint16),
        ("kernelPerPart", C.c_int16),
        ("rowIdx", C.POINTER(C.c_int)),
        ("numInRow", C.POINTER(C.c_int)),
        ("numInRow2", C.POINTER(C.c_int)),
        ("I", C.POINTER(C.c_int)),
        ("J", C.POINTER(C.c_int)),
        ("V", C.POINTER(C.c_double)),
        ("diag", C.POINTER(C.c_double)),
        ("partBoundary", C.POINTER(C.c_int)),
        ("reorderList", C.POINTER(C.c_int)),
    ]


class Config(C.Structure):
    """ehyb_config of include/ehyb.h."""

    _fields_ = [
        ("lds_doubles", C.c_int32),
        ("part_rows", C.c_int32),
        ("threads", C.c_int32),
        ("window_mode", C.c_int32),
        ("items_per_cu", C.c_int32),
        ("partitioner", C.c_int32),
        ("er_seg_len", C.c_int32),
        ("host_threads", C.c_int32),
        ("verbose", C.c_int32),
        ("seed", C.c_int32),
        ("n_top", C.c_int32),
        ("er_threads", C.c_int32),
        ("ell_variant", C.c_int32),
        ("col_sharing", C.c_int32),
        ("fuse_er", C.c_int32),
        ("cap_split", C.c_int32),
        ("hub_rule", C.c_int32),
        ("sym_pairs", C.c_int32),
        ("part_boundary_cap", C.c_int32),
        ("er_mode", C.c_int32),
        ("er_panel_cols", C.c_int32),
        ("er_block_rows", C.c_int32),
        ("direct", C.c_int32),
        ("ell_prune", C.c_int32),
        ("value_map", C.c_int32),
        ("prune_pct", C.c_int32),
        ("er_units1", C.c_int32),
        ("er_units2", C.c_int32),
        ("graph_compress", C.c_int32),
        ("balance", C.c_int32),
        ("req_margin", C.c_int32),
        ("sym_slack_permille", C.c_int32),
        ("xcd_map", C.c_int32),
        ("graphs", C.c_int32),
        ("er_sums", C.c_int32),
        ("er_panel_threads", C.c_int32),
        ("er_queue", C.c_int32),
        ("symbolic", C.c_int32),
        ("cg_fused_dot", C.c_int32),
        ("ell_alternate", C.c_int32),
        ("row_split", C.c_int32),
        ("col_map", C.c_int32),
        ("er_nt", C.c_int32),
        ("ell_nt", C.c_int32),
        ("reserved", C.c_int32 * 21),
    ]


_STAT_NAMES = [
    "nnz", "nnz_ell", "nnz_er", "ell_padding", "size_block_ell", "size_er", "rows_er",
    "er_segments", "n_rows", "n_cols", "n_parts", "n_slabs", "n_items", "halo_cols",
    "window_loads", "bytes_format", "bytes_alg", "max_row", "lds_bytes", "col_words",
    "er_inline", "sym_pairs", "bytes_format_ell", "er_partials",
]


class Stats(C.Structure):
    _fields_ = [(n, C.c_int64) for n in _STAT_NAMES]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n in _STAT_NAMES}


# every symbol include/*.h declares with C linkage: name -> (restype, argtypes)
_P = C.POINTER
_vp = C.c_void_p
_dp = _P(C.c_double)
_ip = _P(C.c_int)
_i64p = _P(C.c_int64)
_cfgp = _P(Config)
_mp = _P(MatrixCOO)
SIGNATURES = {
    # spmv.h
    "spmvGPuEHYB": (None, [_mp, _dp, _dp, C.c_int, _ip]),
    "spmvGPuEHYB_status": (C.c_int, [_mp, _dp, _dp, C.c_int, _ip]),
    # ehyb.h
    "ehyb_last_error": (C.c_char_p, []),
    "ehyb_version": (C.c_char_p, []),
    "ehyb_host_threads": (C.c_int, []),
    "ehyb_config_default": (None, [_cfgp]),
    "ehyb_config_resolve": (None, [_cfgp, _cfgp]),
    "ehyb_sizing": (C.c_int, [C.c_int, _cfgp, _ip, _ip, _ip]),
    "ehyb_partition_graph": (C.c_int, [C.c_int, _i64p, _ip, _ip, C.c_int, C.c_int, _cfgp, _ip, _i64p]),
    "ehyb_matrix_reorder": (C.c_int, [_mp, C.c_int, _cfgp]),
    "ehyb_matrix_reorder_blocks": (C.c_int, [_mp, C.c_int, _cfgp, _ip]),
    "spmvGPuEHYB_cfg": (C.c_int, [_mp, _dp, _dp, C.c_int, _ip, _cfgp, _dp]),
    "ehyb_vector_reorder": (None, [C.c_int, _dp, _dp, _ip]),
    "ehyb_vector_recover": (None, [C.c_int, _dp, _dp, _ip]),
    "ehyb_top_boundary": (C.c_int, [_mp, _cfgp, C.c_int, _ip]),
    "ehyb_plan_create_host": (C.c_int, [_mp, C.c_int, C.c_int, _cfgp, _P(_vp)]),
    "ehyb_plan_create_host_segs": (C.c_int, [_mp, C.c_int, C.c_int, _cfgp, C.c_int, _ip, _P(_vp)]),
    "ehyb_plan_upload": (C.c_int, [_vp]),
    "ehyb_plan_create": (C.c_int, [_mp, _cfgp, _P(_vp)]),
    "ehyb_plan_create_segs": (C.c_int, [_mp, C.c_int, C.c_int, _cfgp, C.c_int, _ip, _P(_vp)]),
    "ehyb_plan_destroy": (None, [_vp]),
    "ehyb_matrix_key": (C.c_uint64, [_mp]),
    "ehyb_plan_save": (C.c_int, [_vp, _ip, C.c_uint64, C.c_char_p]),
    "ehyb_plan_load": (C.c_int, [C.c_char_p, C.c_uint64, _P(_vp), _ip]),
    "ehyb_plan_stats": (C.c_int, [_vp, _P(Stats)]),
    "ehyb_plan_host_array": (C.c_int, [_vp, C.c_int, _P(_vp), _i64p]),
    "ehyb_spmv": (C.c_int, [_vp, _vp, _vp, _vp]),
    "ehyb_spmv_phase": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int]),
    "ehyb_spmv_walk": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int]),
    "ehyb_spmv_graph_create": (C.c_int, [_vp, _vp, _vp, C.c_int, _P(_vp)]),
    "ehyb_graph_launch": (C.c_int, [_vp, _vp]),
    "ehyb_graph_destroy": (None, [_vp]),
    "ehyb_plan_tune": (C.c_int, [_vp, _vp, _vp, C.c_int, _dp, _dp]),
    "ehyb_spmv_part": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int]),
    "ehyb_plan_col_segs": (C.c_int, [_vp, _ip]),
    "ehyb_gather": (C.c_int, [_vp, _vp, _vp, C.c_int64, _vp]),
    "ehyb_scatter_add": (C.c_int, [_vp, _vp, _vp, C.c_int64, _vp]),
    "ehyb_step_pack": (C.c_int, [_vp, _vp, _vp, C.c_int64, _vp, _vp]),
    "ehyb_step_part": (C.c_int, [_vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int]),
    "ehyb_halo_step": (C.c_int, [_vp, _vp, _vp, _vp, _vp, C.c_int64, C.c_int, _vp, _vp, _vp, _vp]),
    "ehyb_rccl_version": (C.c_int, [_ip, C.c_char_p, C.c_int]),
    "ehyb_comm_unique_id": (C.c_int, [_vp]),
    "ehyb_comm_create": (C.c_int, [_vp, C.c_int, C.c_int, _P(_vp)]),
    "ehyb_comm_destroy": (None, [_vp]),
    "ehyb_comm_info": (C.c_int, [_vp, _ip, _ip, _P(_vp)]),
    "ehyb_comm_allreduce_sum": (C.c_int, [_vp, _vp, C.c_int64, _vp]),
    "ehyb_comm_allgather": (C.c_int, [_vp, _vp, _vp, C.c_int64, _vp]),
    "ehyb_halo_create": (C.c_int, [_vp, _vp, C.c_int, _P(C.c_int32), C.c_int64, _i64p, _i64p, _P(_vp)]),
    "ehyb_halo_destroy": (None, [_vp]),
    "ehyb_halo_spmv": (C.c_int, [_vp, _vp, _vp, _vp]),
    "ehyb_halo_set_partials": (C.c_int, [_vp, C.c_int, _i64p, _i64p, _P(C.c_int32), C.c_int64]),
    "ehyb_gather_spmv": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int64, _vp]),
    "ehyb_spmv_bench": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int, C.c_int, _dp, _dp, _dp]),
    "ehyb_spmv_host": (C.c_int, [_vp, _dp, _dp, C.c_int]),
    "ehyb_plan_set_values": (C.c_int, [_vp, _vp, C.c_int64, _vp, C.c_int, _vp]),
    "ehyb_entry_order": (C.c_int, [C.c_int, _ip, _ip, _P(C.c_int32)]),
    "ehyb_device_count": (C.c_int, [_ip]),
    "ehyb_device_set": (C.c_int, [C.c_int]),
    "ehyb_device_name": (C.c_int, [C.c_char_p, C.c_int]),
    "ehyb_dev_alloc": (C.c_int, [C.c_size_t, _P(_vp)]),
    "ehyb_dev_free": (C.c_int, [_vp]),
    "ehyb_h2d": (C.c_int, [_vp, _vp, C.c_size_t]),
    "ehyb_d2h": (C.c_int, [_vp, _vp, C.c_size_t]),
    "ehyb_dev_sync": (C.c_int, []),
    "ehyb_stream_create": (C.c_int, [_P(_vp)]),
    "ehyb_stream_destroy": (C.c_int, [_vp]),
    "ehyb_stream_sync": (C.c_int, [_vp]),
    "ehyb_dev_mem_info": (C.c_int, [_P(C.c_size_t), _P(C.c_size_t)]),
    "ehyb_measure_read_bw": (C.c_int, [C.c_size_t, C.c_int, _dp]),
    "ehyb_cg": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_double, C.c_int, _vp, _ip, _dp]),
    "ehyb_pcg": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int, C.c_double, C.c_int, _vp, _ip, _dp]),
    "ehyb_cg_layout": (C.c_int, [_ip, _ip, _ip, _ip, _ip, _ip]),
    "ehyb_cg_init_step": (C.c_int, [C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ehyb_cg_dot_step": (C.c_int, [C.c_int, _vp, _vp, _vp, _vp]),
    "ehyb_cg_update_step": (C.c_int, [C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int, _vp]),
    "ehyb_cg_direction_step": (C.c_int, [C.c_int, _vp, _vp, _vp, _vp, C.c_int, _vp]),
    "ehyb_mm_read": (C.c_int, [C.c_char_p, _cfgp, _mp, _ip]),
    "ehyb_mm_write": (C.c_int, [C.c_char_p, _mp, C.c_int]),
    "ehyb_matrix_from_csr": (C.c_int, [C.c_int, _i64p, _ip, _dp, _cfgp, _mp]),
    "ehyb_matrix_free": (None, [_mp]),
    "ehyb_x_glibc": (None, [C.c_int, _dp]),
    "ehyb_gen_banded": (C.c_int, [C.c_int, C.c_int, C.c_int, _cfgp, _mp]),
    "ehyb_gen_fem3d": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64, _cfgp, _mp]),
    "ehyb_gen_fem3d_graded": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64, _cfgp, _mp]),
    "ehyb_gen_fem3d_block": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_int, C.c_int,
                                       _cfgp, _mp]),
    "ehyb_matrix_append_ghosts": (C.c_int, [_mp, C.c_int, C.c_int64, _ip, _ip, _dp]),
    "ehyb_matrix_append_rows": (C.c_int, [_mp, C.c_int, C.c_int, C.c_int64, _ip, _ip, _dp, C.c_int]),
    "ehyb_gen_rmat": (C.c_int, [C.c_int, C.c_int64, C.c_uint64, _cfgp, _mp]),
    "ehyb_gen_rmat_block": (C.c_int, [C.c_int, C.c_int64, C.c_uint64, C.c_int, C.c_int, _ip, _cfgp, _mp]),
    "ehyb_gen_rmat_rows": (C.c_int, [C.c_int, C.c_int64, C.c_uint64, C.c_int, C.c_int, _cfgp, _mp]),
    "ehyb_gen_rmat_block_cost": (C.c_int, [C.c_int, C.c_int64, C.c_uint64, C.c_int, C.c_int, C.c_int, _ip, _cfgp, _mp]),
    "ehyb_gen_stencil2d": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64, _cfgp, _mp]),
    "ehyb_gen_kkt3d": (C.c_int, [C.c_int, _cfgp, _mp]),
    "ehyb_gen_mesh3d": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64, _cfgp, _mp]),
}

# C++-linkage names of include/reordering.h, as the reference's driver links them (reordering.h:6-10;
# the reference compiles its .c files as C++, Makefile:6,21-22)
CXX_SIGNATURES = {
    "_Z13matrixReorderP10_matrixCOO": (None, [_mp]),
    "_Z19matrixReorder_unsymP10_matrixCOO": (None, [_mp]),
    "_Z13vectorReorderiPKdPdPKi": (None, [C.c_int, _dp, _dp, _ip]),
    "_Z13vectorRecoveriPKdPdPKi": (None, [C.c_int, _dp, _dp, _ip]),
}

_lib = None


def load():
    """Load libehyb.so (once) and attach argument types.  Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build the HIP library first "
            "(python -c 'import __graft_entry__ as g; g.build()' or make -C ehyb_spmv_gpu_amd/csrc). "
            "There is no CPU fallback for the EHYB multiply."
        )
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here = header/library mismatch
        fn.restype = res
        fn.argtypes = args
    for name, (res, args) in CXX_SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
