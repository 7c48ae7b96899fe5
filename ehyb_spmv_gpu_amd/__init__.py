"""MI355X-native Explicit-Caching HYB fp64 SpMV: Python host mirror over libehyb.so.

The compute path is the HIP library built from csrc/ (see include/ehyb.h for the C-ABI);
importing this package does not load it, `ehyb_spmv_gpu_amd.host` does on first use and
raises if it has not been built.
"""
from . import host  # noqa: F401
from .host import (  # noqa: F401
    ARRAYS,
    EHYB_PART_AUTO,
    EHYB_PART_CONTIGUOUS,
    EHYB_PART_DEGREE,
    EHYB_PART_MTMETIS,
    EHYB_PART_MULTILEVEL,
    EHYB_WINDOW_HALO,
    EHYB_WINDOW_REFERENCE,
    DeviceBuffer,
    EhybError,
    Matrix,
    Plan,
    SpmvGraph,
    Stream,
    device_count,
    entry_order,
    host_threads,
    make_config,
    partition_graph,
    sizing,
    spmv_gpu_ehyb,
    vector_recover,
    vector_reorder,
    x_glibc,
)

__all__ = [n for n in dir() if not n.startswith("_")]
