"""The drop-in boundary as a C-ABI: every symbol the headers declare is exported, the structs
have the reference's layout (spmv.h:17-33), the headers compile as C, and misuse fails loudly.
No GPU compute is called here."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INC = os.path.join(ROOT, "include")


def declared_symbols():
    names = set()
    for h in ("ehyb.h", "spmv.h"):
        text = open(os.path.join(INC, h)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        for m in re.finditer(r"\b(ehyb_\w+|spmvGPuEHYB\w*)\s*\(", text):
            names.add(m.group(1))
    names.discard("ehyb_status")
    return names


def test_every_declared_symbol_is_exported_and_bound():
    from ehyb_spmv_gpu_amd import _lib

    lib = _lib.load()
    declared = declared_symbols()
    assert len(declared) >= 35
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/ but not exported by libehyb.so"
    assert declared == set(_lib.SIGNATURES), "ctypes table and headers disagree"
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r" T (\w+)", out))
    assert declared <= exported
    # the C++-linkage names of reordering.h (the reference's .c files are compiled as C++)
    dem = subprocess.run(["nm", "-D", "-C", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    for sig in ("matrixReorder(_matrixCOO*)", "matrixReorder_unsym(_matrixCOO*)",
                "vectorReorder(int, double const*, double*, int const*)",
                "vectorRecover(int, double const*, double*, int const*)"):
        assert sig in dem, sig
    assert b"gfx950" in lib.ehyb_version()


def test_struct_layout_matches_reference_contract(tmp_path):
    """matrixCOO: 4 x int32, uint16, int16, nine pointers (reference spmv.h:17-33)."""
    from ehyb_spmv_gpu_amd._lib import Config, MatrixCOO, Stats

    src = tmp_path / "layout.c"
    src.write_text(r'''
#include <stdio.h>
#include <stddef.h>
#include "ehyb.h"      /* must compile as C99 */
#include "spmv.h"
int main(void) {
    printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(matrixCOO), offsetof(matrixCOO, totalNum),
           offsetof(matrixCOO, dimension), offsetof(matrixCOO, nParts), offsetof(matrixCOO, vectorCacheSize),
           offsetof(matrixCOO, kernelPerPart), offsetof(matrixCOO, rowIdx), offsetof(matrixCOO, V),
           offsetof(matrixCOO, reorderList));
    printf("%zu %zu %zu\n", sizeof(ehyb_config), sizeof(ehyb_stats), sizeof(cb_s));
    cb_s cb; init_cb(&cb);
    printf("%d %d %d %d %d\n", cb.PRECOND, cb.GPU, cb.RODR, cb.BLOCK, cb.CACHE);
    return 0;
}
''')
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", INC, str(src), "-o", str(exe)], check=True)
    l1, l2, l3 = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.splitlines()
    size, o_total, o_dim, o_nparts, o_cache, o_kpp, o_rowidx, o_v, o_list = map(int, l1.split())
    assert (o_total, o_dim, o_nparts, o_cache, o_kpp) == (0, 4, 12, 16, 18)
    assert o_rowidx == 24 and o_v == 24 + 5 * 8 and o_list == 24 + 8 * 8 and size == 96
    assert size == C.sizeof(MatrixCOO) and o_rowidx == MatrixCOO.rowIdx.offset and o_list == MatrixCOO.reorderList.offset
    cfg_size, stats_size, cb_size = map(int, l2.split())
    assert cfg_size == C.sizeof(Config) and stats_size == C.sizeof(Stats) and cb_size == 7
    assert l3.split() == ["0", "0", "1", "1", "1"]  # init_cb defaults, spmv.h:65-73


def test_headers_compile_as_cpp_and_link(tmp_path):
    """A C++ caller in the style of solver_test.c links against the C++ names of reordering.h."""
    src = tmp_path / "caller.cpp"
    src.write_text(r'''
#include <stdlib.h>
#include "spmv.h"
#include "reordering.h"
#include "ehyb.h"
int main() {
    double a[3] = {1, 2, 3}, b[3], c[3];
    int list[3] = {2, 0, 1};
    vectorReorder(3, a, b, list);
    vectorRecover(3, b, c, list);
    void (*fn)(matrixCOO*, const double*, double*, const int, int*) = &spmvGPuEHYB;
    void (*r1)(matrixCOO*) = &matrixReorder; void (*r2)(matrixCOO*) = &matrixReorder_unsym;
    return (c[0] == 1 && c[1] == 2 && c[2] == 3 && b[2] == 1 && fn && r1 && r2) ? 0 : 1;
}
''')
    lib_dir = os.path.join(ROOT, "ehyb_spmv_gpu_amd")
    exe = tmp_path / "caller"
    subprocess.run(["g++", "-I", INC, str(src), "-o", str(exe), "-L", lib_dir, "-lehyb",
                    f"-Wl,-rpath,{lib_dir}", "-Wl,-rpath,/opt/rocm/lib"], check=True)
    assert subprocess.run([str(exe)]).returncode == 0


def test_argument_errors_are_reported(E):
    from ehyb_spmv_gpu_amd import _lib

    lib = _lib.load()
    assert lib.ehyb_plan_create_host(None, 0, 0, None, None) == 1          # EHYB_ERR_ARG
    h = C.c_void_p()
    assert lib.ehyb_plan_create_host(None, 0, 0, None, C.byref(h)) == 1
    assert b"null" in lib.ehyb_last_error()
    assert lib.ehyb_spmv(None, None, None, None) == 1
    assert lib.ehyb_plan_stats(None, None) == 1
    assert lib.ehyb_sizing(-5, None, None, None, None) == 1
    assert lib.spmvGPuEHYB_status(None, None, None, 1, None) == 1
    m = E.Matrix.generate("stencil2d", 8, 8, 5, 0, 1)
    plan = E.Plan(m, upload=False)
    with pytest.raises(E.EhybError) as ei:
        plan.array("ell_val")  # fine
        E.host._check(lib.ehyb_plan_host_array(plan.h, 99, C.byref(C.c_void_p()), C.byref(C.c_int64())), "host_array")
    assert ei.value.code == 1


def test_no_cpu_fallback_for_the_multiply(E):
    """Without a device the product path fails loudly instead of computing on the host."""
    if E.device_count() > 0:
        pytest.skip("a GPU is visible")
    m = E.Matrix.generate("stencil2d", 8, 8, 5, 0, 1)
    m.reorder()
    plan = E.Plan(m, upload=False)
    with pytest.raises(E.EhybError) as ei:
        plan.upload()
    assert ei.value.code == 4 and "no CPU fallback" in str(ei.value)       # EHYB_ERR_NO_DEVICE
    x = np.ones(m.n)
    with pytest.raises(E.EhybError) as ei:
        plan.spmv_host(x)
    assert ei.value.code == 8                                              # EHYB_ERR_STATE: not uploaded
    with pytest.raises(E.EhybError):
        E.spmv_gpu_ehyb(m, x, 1)


def test_product_does_not_touch_the_oracle():
    """Nothing under the package imports, links or executes anything under oracle/."""
    pkg = os.path.join(ROOT, "ehyb_spmv_gpu_amd")
    for base, _, files in os.walk(pkg):
        if os.sep + "build" in base or "__pycache__" in base:
            continue
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", "Makefile")):
                text = open(os.path.join(base, f), errors="replace").read()
                assert "oracle" not in text.lower(), f"{f} mentions the oracle"
    out = subprocess.run(["ldd", os.path.join(pkg, "libehyb.so")], capture_output=True, text=True).stdout
    assert "oracle" not in out


_ENV_KNOBS = {"EHYB_PB_PROBE": "3", "EHYB_PRUNE_PCT": "50", "EHYB_PB_UNITS1": "7", "EHYB_PB_UNITS2": "5", "EHYB_COMPRESS": "0",
              "EHYB_FORCE_WEIGHTED": "1", "EHYB_REQ_MARGIN": "9", "EHYB_SYM_SLACK_PERMILLE": "200", "EHYB_XCD_MAP": "0",
              "EHYB_BENCH_GRAPH": "0", "EHYB_CG_GRAPH": "0", "EHYB_ER_SUMS": "2", "EHYB_PARTITIONER": "1"}

_DIGEST_SNIPPET = r"""
import hashlib, sys
sys.path.insert(0, %r)
import ehyb_spmv_gpu_amd as E
h = hashlib.sha256()
for gen, kw in ((("fem3d", 60000, 3, 28, 28, 13500, 1, 1), dict(sym_pairs=1)), (("rmat", 16, 1 << 19, 3), dict(lds_doubles=2048, er_mode=2))):
    cfg = E.make_config(**kw)
    m = E.Matrix.generate(*gen, cfg=cfg)
    m.reorder(cfg)
    h.update(m.reorder_list.tobytes())
    plan = E.Plan(m, cfg, upload=False)
    for name in sorted(E.host.ARRAYS):
        h.update(plan.array(name).tobytes())
print(h.hexdigest())
"""


def test_no_tuning_variable_is_read_from_the_environment():
    """Every tuning knob is an ehyb_config field: the only getenv left in the library names the optional mt-metis
    shared object, and the permutation and every array of a plan come out the same whatever EHYB_* variables
    the process carries (round 2 read twelve of them; one -- EHYB_PB_PROBE -- made the multiply wrong on purpose)."""
    import re
    import sys

    src = os.path.join(ROOT, "ehyb_spmv_gpu_amd", "csrc")
    hits = []
    for f in sorted(os.listdir(src)):
        if f.endswith((".cpp", ".hip", ".h")):
            hits += [(f, m) for m in re.findall(r'getenv\("([A-Z_]+)"\)', open(os.path.join(src, f)).read())]
    assert hits == [("reorder.cpp", "EHYB_MTMETIS_LIB")], hits
    code = _DIGEST_SNIPPET % ROOT
    clean = {k: v for k, v in os.environ.items() if not k.startswith("EHYB_")}
    a = subprocess.run([sys.executable, "-c", code], env=clean, capture_output=True, text=True, timeout=600)
    b = subprocess.run([sys.executable, "-c", code], env=dict(clean, **_ENV_KNOBS), capture_output=True, text=True, timeout=600)
    assert a.returncode == 0 and b.returncode == 0, a.stderr[-800:] + b.stderr[-800:]
    assert a.stdout.strip() == b.stdout.strip() and len(a.stdout.strip()) == 64


@pytest.mark.gpu
def test_multiply_ignores_the_environment(E, O, gpu, monkeypatch):
    """The panel-form multiply with round 2's probe variable (and the rest) set in the environment: still right."""
    for k, v in _ENV_KNOBS.items():
        monkeypatch.setenv(k, v)
    cfg = E.make_config(lds_doubles=2048, er_mode=2, fuse_er=2)
    m = E.Matrix.generate("rmat", 16, 1 << 19, 3, cfg=cfg)
    x = O.x_glibc(m.n)
    y_ref = O.spmv_coo(m.n, m.I, m.J, m.V, x)
    scale = O.abs_rowsum(m.n, m.I, m.J, m.V, x)
    m.reorder(cfg)
    plan = E.Plan(m, cfg)
    assert plan.stats["er_partials"] > 0
    y = E.vector_recover(plan.spmv_host(E.vector_reorder(x, m.reorder_list)), m.reorder_list)
    bad, worst = O.check_tolerance(y, y_ref, scale)
    assert bad == 0, worst
