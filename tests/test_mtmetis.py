"""The mt-metis backend of the reorder step (cfg.partitioner = EHYB_PART_MTMETIS; reordering.c:116-139,
270-293 call MTMETIS_PartGraphKway) run for real: oracle/_ref/mtmetis_driver links the reference's
vendored libmtmetis.a statically -- as the reference's own Makefile does -- and libehyb.so finds the
partitioner in the process image.  Reorder -> layout -> oracle walk -> reference product, and the
partition quality next to the built-in partitioner's (the table in DESIGN.md comes from the same tool).
Skipped where neither the reference tree nor a prebuilt driver exists."""
import json
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER = os.path.join(ROOT, "oracle", "_ref", "mtmetis_driver")

CASES = [
    ("stencil2d", (150, 150, 5, 3000, 1), 1024),
    ("fem3d", (30000, 3, 22, 22, 13500, 1, 1), 4096),
    ("rmat", (14, 1 << 17, 1), 2048),
]


@pytest.mark.parametrize("kind,args,lds", CASES, ids=[c[0] for c in CASES])
def test_mtmetis_backend_end_to_end(E, O, tmp_path, kind, args, lds):
    if os.path.isdir("/root/reference"):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "ref"], check=True)
    if not os.path.exists(DRIVER):
        pytest.skip("oracle/_ref/mtmetis_driver not built (needs /root/reference/libmtmetis.a at build time)")
    prefix = str(tmp_path / "mt")
    p = subprocess.run([DRIVER, kind] + [str(a) for a in args] + ["--", prefix, str(lds)], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout + p.stderr
    builtin, mt = [json.loads(l) for l in p.stdout.strip().splitlines() if l.startswith("{")]
    assert "mt-metis" in mt["partitioner"] and "built-in" in builtin["partitioner"]
    assert mt["parts"] == builtin["parts"] and mt["rows"] == builtin["rows"]
    # same ball park (mt-metis runs on one thread here, like the reference's symmetric call)
    assert mt["cut_entries"] <= 1.5 * builtin["cut_entries"] + 100 and builtin["cut_entries"] <= 1.5 * mt["cut_entries"] + 100
    assert abs(mt["ell_share"] - builtin["ell_share"]) < 0.05
    # the plan built on mt-metis' partition, walked by the oracle, gives the reference product
    m = E.Matrix.generate(kind, *args, cfg=E.make_config(lds_doubles=lds))
    x = O.x_glibc(m.n)
    y_ref = O.spmv_coo(m.n, m.I, m.J, m.V, x)
    scale = O.abs_rowsum(m.n, m.I, m.J, m.V, x)
    plan, perm = E.Plan.load(prefix + ".plan", key=m.key(), upload=False)
    assert sorted(perm.tolist()) == list(range(m.n))
    y, written = O.walk_plan(plan, E.vector_reorder(x, perm))
    assert written.min() == 1
    bad, worst = O.check_tolerance(E.vector_recover(y, perm), y_ref, scale)
    assert bad == 0, f"worst {worst:.3e}"


def test_mtmetis_backend_reports_its_absence(E):
    """In a process that does not contain mt-metis the backend refuses (status 8) instead of silently
    partitioning some other way."""
    if os.environ.get("EHYB_MTMETIS_LIB"):
        pytest.skip("EHYB_MTMETIS_LIB is set")
    cfg = E.make_config(partitioner=E.EHYB_PART_MTMETIS, lds_doubles=1024)
    m = E.Matrix.generate("stencil2d", 60, 60, 5, 100, 1, cfg=cfg)
    with pytest.raises(E.EhybError) as e:
        m.reorder(cfg)
    assert e.value.code == 8 and "mt-metis" in str(e.value)
