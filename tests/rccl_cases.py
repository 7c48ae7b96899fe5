"""The RCCL-native step (csrc/ehyb_comm.hip: ehyb_comm_*, ehyb_halo_spmv) on the GPU box's ONE device: a communicator with a
single rank whose ghost columns are some of its OWN columns (RankLocalMatrix(loopback=...)), so that pack -> grouped
ncclSend / ncclRecv to itself -> ghost columns -> the multiply in parts all run through RCCL's real enqueue path.  Checked
against the plain step on the same plan (a device copy in place of the collective, then one multiply) and against the CPU
oracle.  The N > 1 logic of the send lists is covered over gloo (tests/test_distributed_cpu.py, tests/test_gpu_dist.py).

Run by tests/test_gpu_rccl.py in a process of its own that imports torch BEFORE libehyb.so is loaded: torch's wheel brings its
own libamdhip64.so, and RCCL (torch's copy) must see the HIP runtime libehyb.so uses -- the order bench.py has."""
import ctypes as C
import os
import sys
import time

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu

import torch  # noqa: E402,F401  (before anything loads libehyb.so)


@pytest.fixture(scope="module")
def comm(gpu):
    import torch  # noqa: F401  (first: libehyb.so then opens the RCCL torch has loaded -- one copy per process)

    from ehyb_spmv_gpu_amd import dist as D

    c = D.make_comm()            # no process group: one rank, the unique id never leaves the process
    assert c.world == 1 and c.rank == 0 and c.stream
    yield c
    c.destroy()


def _case(E, O, comm, gen, gargs, cfg_gen, cfg_plan, chunks, shares, loopback, symmetric=False, exchange="halo"):
    import torch

    from ehyb_spmv_gpu_amd import dist as D

    dev = torch.device("cuda", 0)
    m = E.Matrix.generate(gen, *gargs, cfg=cfg_gen)
    n = m.n
    I, J, V = m.I.copy(), m.J.copy(), m.V.copy()
    m.free()
    x = O.x_glibc(n)
    y_ref = O.spmv_coo(n, I, J, V, x)
    scale = O.abs_rowsum(n, I, J, V, x)
    L = D.RankLocalMatrix(I, J, V, [0, n], 0, cfg_plan, symmetric=symmetric, chunks=chunks, chunk_shares=shares, loopback=loopback, exchange=exchange)
    assert L.exchanges and L.n_ghost > 0 and int(L.send_counts.sum()) == int(L.recv_counts.sum()) == L.n_ghost
    plain = D.HaloSpmv(L, dev, overlap=False)                 # the reference step: device copies, then ehyb_spmv
    rccl = D.HaloSpmv(L, dev, comm=comm)                      # ONE C call per step, the exchange through RCCL
    assert rccl.c_halo is not None
    return L, plain, rccl, x, y_ref, scale


def _run(sh, x, steps=1):
    import torch

    sh.set_x_local(x)
    for s in range(steps):
        sh.y.fill_(float("nan"))
        sh.x[sh.L.n_loc:].fill_(float("nan"))       # the ghost columns must come from THIS step's exchange
        sh.step()
    torch.cuda.synchronize()
    return sh.y_local()


def test_rccl_is_the_one_the_process_already_holds(E, comm):
    from ehyb_spmv_gpu_amd import _lib

    v, where = C.c_int(), C.create_string_buffer(256)
    assert _lib.load().ehyb_rccl_version(C.byref(v), where, 256) == 0 and v.value >= 21800
    loaded = {ln.split()[-1] for ln in open("/proc/self/maps") if "librccl" in ln}
    assert len(loaded) == 1, loaded           # torch's and libehyb's RCCL are the same mapped file


def test_loopback_step_fem_windows_bit_for_bit(E, O, comm):
    """Windows kept, CSR residual over the ghost columns: deterministic kernels -> the RCCL step equals the plain step bit for bit."""
    cfg = E.make_config(lds_doubles=4096)
    L, plain, rccl, x, y_ref, scale = _case(E, O, comm, "fem3d", (30000, 3, 22, 22, 13500, 0, 7), cfg, cfg, 2, None, 0.3)
    y0, y1 = _run(plain, x), _run(rccl, x, steps=3)
    assert O.check_tolerance(y1, y_ref, scale)[0] == 0
    assert np.array_equal(y0, y1)


@pytest.mark.parametrize("chunks,shares", [(1, None), (3, [0.2, 0.3, 0.5])])
def test_loopback_step_rmat_panel_form(E, O, comm, chunks, shares):
    """R-MAT in panel form (what bench.py --gpus N runs): every chunk's panels multiply behind its own ncclRecv."""
    cfg = E.make_config(partitioner=E.EHYB_PART_DEGREE)
    cfgp = E.make_config(partitioner=E.EHYB_PART_DEGREE, er_mode=2, er_panel_cols=4096)
    L, plain, rccl, x, y_ref, scale = _case(E, O, comm, "rmat", (17, 1 << 20, 1), cfg, cfgp, chunks, shares, 0.6)
    assert rccl.plan.stats["er_partials"] > 0 and rccl.plan.col_segs == chunks + 1
    y0, y1 = _run(plain, x), _run(rccl, x, steps=2)
    assert O.check_tolerance(y0, y_ref, scale)[0] == 0 and O.check_tolerance(y1, y_ref, scale)[0] == 0
    # same launches on the same data: only the order of pass 2's LDS adds may differ
    assert float(np.max(np.abs(y0 - y1) / np.maximum(scale, 1e-300))) < 1e-14


def test_loopback_step_follows_a_changing_x(E, O, comm):
    """A solver's loop: x changes between steps, and step k+1's exchange overwrites the ghost columns step k's multiply read."""
    import torch

    cfg = E.make_config(lds_doubles=4096)
    L, plain, rccl, x, y_ref, scale = _case(E, O, comm, "fem3d", (24000, 3, 20, 20, 13500, 0, 3), cfg, cfg, 2, [0.5, 0.5], 0.4)
    rccl.set_x_local(x)
    ys = []
    for k in range(6):                       # no synchronisation between the steps
        rccl.step()
        ys.append(rccl.y.clone())
        rccl.x[:L.n_loc].mul_(-0.5)          # own entries only: the ghosts must follow through the exchange
    torch.cuda.synchronize()
    for k, y in enumerate(ys):
        got = L.y_from_plan(y.cpu().numpy())
        assert O.check_tolerance(got, y_ref * (-0.5) ** k, scale * 0.5 ** k)[0] == 0, k


def test_comm_collectives_single_rank(E, comm):
    import torch

    dev = torch.device("cuda", 0)
    a = torch.arange(1000, dtype=torch.float64, device=dev)
    comm.allreduce_sum(a.data_ptr(), 1000, torch.cuda.current_stream().cuda_stream)
    b = torch.zeros(1000, dtype=torch.float64, device=dev)
    from ehyb_spmv_gpu_amd import _lib

    assert _lib.load().ehyb_comm_allgather(comm.h, a.data_ptr(), b.data_ptr(), 1000, torch.cuda.current_stream().cuda_stream) == 0
    torch.cuda.synchronize()
    assert torch.equal(a.cpu(), torch.arange(1000, dtype=torch.float64)) and torch.equal(a, b)


def test_halo_create_checks_its_lists(E, O, comm):
    from ehyb_spmv_gpu_amd import _lib

    cfg = E.make_config(lds_doubles=4096)
    L, plain, rccl, *_ = _case(E, O, comm, "fem3d", (24000, 3, 20, 20, 13500, 0, 3), cfg, cfg, 2, None, 0.4)
    lib = _lib.load()
    idx = np.ascontiguousarray(L.send_idx, dtype=np.int32)
    sc = np.ascontiguousarray(L.send_counts, dtype=np.int64).reshape(-1)
    rc = np.ascontiguousarray(L.recv_counts, dtype=np.int64).reshape(-1)
    h = C.c_void_p()
    p32, p64 = C.POINTER(C.c_int32), C.POINTER(C.c_int64)

    def create(chunks, idx, sc, rc):
        return lib.ehyb_halo_create(comm.h, plain.plan.h, chunks, idx.ctypes.data_as(p32), len(idx), sc.ctypes.data_as(p64), rc.ctypes.data_as(p64), C.byref(h))

    assert create(3, idx, sc, rc) == 1                       # the plan has 1 + 2 column segments
    assert create(2, idx, sc + 1, rc) == 1                   # counts that do not add up to the send list
    assert create(2, idx, sc, rc * 2) == 1                   # more than the chunk's column segment holds
    bad = idx.copy()
    bad[0] = L.n_loc + 5
    assert create(2, bad, sc, rc) == 1                       # a ghost column in the send list
    assert b"own" in lib.ehyb_last_error()


def test_host_cost_of_the_c_step(E, O, comm):
    """The point of the C step: the host issues one call per multiply.  (tools/host_step_cost.py prints the numbers; here a
    loose bound so that a regression to per-part Python calls shows.)"""
    import torch

    cfg = E.make_config(partitioner=E.EHYB_PART_DEGREE)
    cfgp = E.make_config(partitioner=E.EHYB_PART_DEGREE, er_mode=2, er_panel_cols=4096)
    L, plain, rccl, x, *_ = _case(E, O, comm, "rmat", (17, 1 << 20, 1), cfg, cfgp, 2, [0.25, 0.75], 0.6)
    rccl.set_x_local(x)
    for _ in range(20):
        rccl.step()
    torch.cuda.synchronize()
    t_issue, done = 0.0, 0
    while done < 240:                # batches on an idle device: a full queue would make the host wait for the device
        t0 = time.perf_counter()
        for _ in range(8):
            rccl.step()
        t_issue += time.perf_counter() - t0
        torch.cuda.synchronize()
        done += 8
    host_us = t_issue / done * 1e6
    print(f"host_us_per_step (C step, 2 chunks, world 1): {host_us:.1f}")
    assert host_us < 100


def test_cover_exchange_over_rccl(E, O, comm):
    """exchange "cover" through the C step (ehyb_halo_set_partials): x chunks out at once, own columns incl. the foreign rows, the
    foreign rows closed and shipped (to the rank itself here), chunks, own rows closed, received partial sums added -- against the
    plain step and the oracle, over several steps with x changing."""
    import torch

    cfg = E.make_config(partitioner=E.EHYB_PART_DEGREE)
    cfgp = E.make_config(partitioner=E.EHYB_PART_DEGREE, er_panel_cols=4096)
    L, plain, rccl, x, y_ref, scale = _case(E, O, comm, "rmat", (17, 1 << 20, 1), cfg, cfgp, 2, [0.25, 0.75], 0.6, exchange="cover")
    assert L.cover and L.n_foreign > 0 and rccl.plan.stats["nnz_ell"] == 0
    print(f"cover: {L.n_ghost} ghost columns + {int(L.yrecv_counts.sum())} partial sums per step, {L.nnz_exported} entries handed over")
    y0 = _run(plain, x)
    assert O.check_tolerance(y0, y_ref, scale)[0] == 0
    rccl.set_x_local(x)
    ys = []
    for k in range(5):
        rccl.x[L.n_loc:].fill_(float("nan"))
        rccl.step()
        ys.append(rccl.y[:L.n_loc].clone())
        rccl.x[:L.n_loc].mul_(-0.5)
    torch.cuda.synchronize()
    for k, y in enumerate(ys):
        assert O.check_tolerance(L.y_from_plan(y.cpu().numpy()), y_ref * (-0.5) ** k, scale * 0.5 ** k)[0] == 0, k


def test_gather_spmv_single_rank(E, O, comm):
    """ehyb_gather_spmv (the all-gather arm as one call): x = [own segment | the segment of every rank as ncclAllGather delivers it].
    One rank: the gathered part is a copy of the own segment -- a matrix whose off-diagonal half reads its columns THERE must
    multiply like the original."""
    import torch

    from ehyb_spmv_gpu_amd import _lib

    cfg = E.make_config(lds_doubles=4096)
    m0 = E.Matrix.generate("fem3d", 24000, 3, 20, 20, 13500, 0, 3, cfg=cfg)
    n = m0.n
    I, J, V = m0.I.copy(), m0.J.copy(), m0.V.copy()
    m0.free()
    x = O.x_glibc(n)
    y_ref = O.spmv_coo(n, I, J, V, x)
    scale = O.abs_rowsum(n, I, J, V, x)
    far = np.abs(I.astype(np.int64) - J) > 40              # these entries read x from the gathered copy
    keep = ~far
    indptr = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(np.bincount(I[keep], minlength=n), out=indptr[1:])
    cfg1 = E.make_config(lds_doubles=4096, n_top=1)
    m = E.Matrix.from_csr(indptr, J[keep], V[keep], cfg1)
    m.reorder(cfg1)
    perm = m.reorder_list[:n].copy()
    seg = n                                                # one rank: its padded segment is the segment
    m.append_ghosts(seg, perm[I[far]], perm[J[far]], V[far])   # ghost column g = place of x entry g in the gathered copy (owner's plan order)
    plan = E.Plan(m, E.make_config(lds_doubles=4096, n_top=2), rows=(0, n))
    dev = torch.device("cuda", 0)
    xd = torch.zeros(2 * seg, dtype=torch.float64, device=dev)
    xd[:n] = torch.from_numpy(E.vector_reorder(x, perm)).to(dev)
    xd[seg:].fill_(float("nan"))                           # must come from the collective
    yd = torch.full((n,), float("nan"), dtype=torch.float64, device=dev)
    lib = _lib.load()
    for _ in range(3):
        assert lib.ehyb_gather_spmv(comm.h, plan.h, xd.data_ptr(), yd.data_ptr(), seg, torch.cuda.current_stream().cuda_stream) == 0, lib.ehyb_last_error()
    torch.cuda.synchronize()
    assert O.check_tolerance(E.vector_recover(yd.cpu().numpy(), perm), y_ref, scale)[0] == 0
