"""Loader for oracle/liboracle.so plus numpy helpers.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module
(see oracle/ehyb_oracle.c for the rules and the parity status).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liboracle.so")
_lib = None


def build():
    subprocess.run(["make", "-C", _HERE, "-s"], check=True)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            build()
        L = C.CDLL(_LIB)
        dp, ip, i64p = C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_int64)
        L.oracle_x_glibc.argtypes = [C.c_int, dp]
        L.oracle_spmv_coo.argtypes = [C.c_int64, ip, ip, dp, dp, dp]
        L.oracle_spmv_sym_lower.argtypes = [C.c_int64, ip, ip, dp, dp, dp]
        L.oracle_spmv_csr.argtypes = [C.c_int, i64p, ip, dp, dp, dp]
        L.oracle_spmv_csr_omp.argtypes = [C.c_int, i64p, ip, dp, dp, dp]
        L.oracle_max_threads.restype = C.c_int
        L.oracle_abs_rowsum.argtypes = [C.c_int64, ip, ip, dp, dp, dp]
        L.oracle_compare.argtypes = [dp, dp, C.c_double, C.c_int, dp, dp]
        L.oracle_compare.restype = C.c_int64
        L.oracle_check_tolerance.argtypes = [dp, dp, dp, C.c_int, C.c_double, dp]
        L.oracle_check_tolerance.restype = C.c_int64
        L.oracle_time_spmv.argtypes = [C.c_int, C.c_int, C.c_int64, i64p, ip, ip, dp, dp, dp, C.c_int]
        L.oracle_time_spmv.restype = C.c_double
        _lib = L
    return _lib


def _d(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _i(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


def _l(a):
    return a.ctypes.data_as(C.POINTER(C.c_int64))


def x_glibc(n):
    x = np.empty(n, dtype=np.float64)
    lib().oracle_x_glibc(n, _d(x))
    return x


def spmv_coo(n, I, J, V, x):
    """The reference CPU product (solver_test.c:102): entries in storage order."""
    I = np.ascontiguousarray(I, dtype=np.int32)
    J = np.ascontiguousarray(J, dtype=np.int32)
    V = np.ascontiguousarray(V, dtype=np.float64)
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.zeros(n, dtype=np.float64)
    lib().oracle_spmv_coo(len(V), _i(I), _i(J), _d(V), _d(x), _d(y))
    return y


def spmv_sym_lower(n, I, J, V, x):
    I = np.ascontiguousarray(I, dtype=np.int32)
    J = np.ascontiguousarray(J, dtype=np.int32)
    V = np.ascontiguousarray(V, dtype=np.float64)
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.zeros(n, dtype=np.float64)
    lib().oracle_spmv_sym_lower(len(V), _i(I), _i(J), _d(V), _d(x), _d(y))
    return y


def spmv_csr(rowptr, col, val, x, omp=False):
    rowptr = np.ascontiguousarray(rowptr, dtype=np.int64)
    col = np.ascontiguousarray(col, dtype=np.int32)
    val = np.ascontiguousarray(val, dtype=np.float64)
    x = np.ascontiguousarray(x, dtype=np.float64)
    n = len(rowptr) - 1
    y = np.zeros(n, dtype=np.float64)
    fn = lib().oracle_spmv_csr_omp if omp else lib().oracle_spmv_csr
    fn(n, _l(rowptr), _i(col), _d(val), _d(x), _d(y))
    return y


def abs_rowsum(n, I, J, V, x):
    I = np.ascontiguousarray(I, dtype=np.int32)
    J = np.ascontiguousarray(J, dtype=np.int32)
    V = np.ascontiguousarray(V, dtype=np.float64)
    x = np.ascontiguousarray(x, dtype=np.float64)
    s = np.zeros(n, dtype=np.float64)
    lib().oracle_abs_rowsum(len(V), _i(I), _i(J), _d(V), _d(x), _d(s))
    return s


def compare(y_result, y, threshold=0.01):
    """compare() of solver_test.c:7-29 -> (offenders, diff, ampldiff)."""
    a = np.ascontiguousarray(y_result, dtype=np.float64)
    b = np.ascontiguousarray(y, dtype=np.float64)
    d, ad = C.c_double(), C.c_double()
    bad = lib().oracle_compare(_d(a), _d(b), threshold, len(a), C.byref(d), C.byref(ad))
    return int(bad), d.value, ad.value


TOLERANCE = 1e-12  # |y_gpu - y_cpu| <= 1e-12 * sum_j |a_ij x_j|  


def check_tolerance(a, b, scale, tol=TOLERANCE):
    """-> (rows violating |a-b| <= tol*scale, worst |a-b|/scale)."""
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    s = np.ascontiguousarray(scale, dtype=np.float64)
    w = C.c_double()
    bad = lib().oracle_check_tolerance(_d(a), _d(b), _d(s), len(a), tol, C.byref(w))
    return int(bad), w.value


def time_spmv(kind, rowptr, I, J, V, x, reps=3):
    """Seconds per multiply (best of reps). kind 0: COO order 1 thread; 1: CSR 1 thread; 2: CSR all cores."""
    rowptr = np.ascontiguousarray(rowptr, dtype=np.int64)
    I = np.ascontiguousarray(I, dtype=np.int32)
    J = np.ascontiguousarray(J, dtype=np.int32)
    V = np.ascontiguousarray(V, dtype=np.float64)
    x = np.ascontiguousarray(x, dtype=np.float64)
    n = len(rowptr) - 1
    y = np.zeros(n, dtype=np.float64)
    t = lib().oracle_time_spmv(kind, n, len(V), _l(rowptr), _i(I), _i(J), _d(V), _d(x), _d(y), reps)
    return t, y


def max_threads():
    return int(lib().oracle_max_threads())


def set_threads(n):
    """OpenMP threads of the timed CSR product (bench.py: the CPUs the process owns, not the ones it sees)."""
    lib().oracle_set_threads(int(n))


# --------------------------------------------------------------------------------------
# CPU walk of the MI355X layout (include/ehyb.h EHYB_ARR_*), indexing the arrays exactly as
# ehyb_ell_kernel / ehyb_er_kernel do.  Checks the host builder without a GPU.
def walk_plan(plan, x):
    x = np.asarray(x, dtype=np.float64)
    n = plan.n
    y = np.zeros(n, dtype=np.float64)
    pb = plan.array("part_boundary")
    win_len = plan.array("win_len")
    halo_ptr = plan.array("halo_ptr")
    halo_cols = plan.array("halo_cols")
    spp = plan.array("slab_pair_ptr").astype(np.int64)
    slab_row = plan.array("slab_row")
    ell_val = plan.array("ell_val")
    ell_col = plan.array("ell_col").astype(np.int64)
    scp = plan.array("slab_col_ptr").astype(np.int64)
    lane_byte = plan.array("lane_group").astype(np.int64).reshape(-1, 64)
    lane_group = lane_byte & 0x3F      # bits 6-7: the lane's part in its group's mirror sum (symmetric pairs only)
    lane_code = lane_byte >> 6
    meta = plan.array("slab_meta").astype(np.int64).reshape(-1, 4)
    items = plan.array("items").reshape(-1, 8)
    segs = plan.array("segs").reshape(-1, 8)
    slab_part = plan.array("slab_part")
    written = np.zeros(n, dtype=np.int32)
    seg_ptr = plan.array("er_seg_ptr")
    seg_row = plan.array("er_seg_row")
    seg_done = np.zeros(len(seg_row), dtype=np.int32)
    inline = plan.stats["er_inline"] > 0
    y_inl = np.zeros(n)
    lrow_tab = plan.array("slab_lrow").astype(np.int64).reshape(-1, 64)
    sym = len(lrow_tab) > 0          # symmetric pair storage: lanes name their rows, one workgroup per partition
    assert sym or plan.stats["sym_pairs"] == 0
    y_mirror = np.zeros(n)
    next_seg, next_slab = 0, 0
    # panel residual whose second pass ASSIGNS the rows of partitions without a window (units with rows < 0):
    # the ELL launch skips those partitions, pass 2 is the only writer of their rows
    u2_all = plan.array("pb_units2").reshape(-1, 4)
    pb_assign = plan.stats["er_partials"] > 0 and bool(np.any(u2_all[:, 3] < 0))
    for g0, g1, is0, is1, e0, e64, e16, e1 in items:
        # an item = consecutive segments covering the consecutive slabs [is0, is1); items follow each other
        # in slab order -- except with symmetric pairs, where an item is a partition and the items are
        # sorted heaviest first (every slab still belongs to exactly one item: `written` checks it)
        assert g0 == next_seg and (g1 > g0 or pb_assign) and is1 > is0 and (sym or is0 == next_slab)
        if pb_assign:
            # partitions without a window have no segment: the item's segments lie inside its slab range, in
            # order, and every slab of the range that no segment covers belongs to such a partition
            cov = np.zeros(is1 - is0, dtype=bool)
            for g in range(g0, g1):
                assert is0 <= segs[g, 1] < segs[g, 2] <= is1 and (g == g0 or segs[g, 1] >= segs[g - 1, 2])
                cov[segs[g, 1] - is0:segs[g, 2] - is0] = True
            assert np.all(win_len[slab_part[is0:is1][~cov]] == 0) and np.all(win_len[slab_part[is0:is1][cov]] > 0)
        else:
            assert segs[g0, 1] == is0 and segs[g1 - 1, 2] == is1
        next_seg, next_slab = g1, is1
        # the item's residual segments: rows inside the item's slab range, bins by length
        assert e0 <= e64 <= e16 <= e1
        if e1 > e0:
            lens = np.diff(seg_ptr[e0:e1 + 1])
            assert np.all(lens[:e64 - e0] >= 128) and np.all((lens[e64 - e0:e16 - e0] > 16) & (lens[e64 - e0:e16 - e0] < 128))
            assert np.all(lens[e16 - e0:] <= 16) and np.all(lens >= 0)
            rows = seg_row[e0:e1] & 0x7FFFFFFF
            last_p = slab_part[is1 - 1]   # (= the partition of the item's last segment, where it has segments)
            assert rows.min() >= slab_row[is0] and rows.max() < min(int(slab_row[is1 - 1]) + 64, int(pb[last_p + 1]))
            seg_done[e0:e1] += 1
        for g in range(g0, g1):
            p, s0, s1, hn, ps, pe, wl, hb = (int(v) for v in segs[g])
            if g > g0:
                assert (s0 == segs[g - 1, 2] or pb_assign) and p > segs[g - 1, 0], "segments of an item are consecutive and cut at partition boundaries"
            assert np.all(slab_part[s0:s1] == p) and s1 > s0
            # the partition scalars carried by the segment record are the partition arrays' values
            assert (ps, pe, wl, hb, hn) == (int(pb[p]), int(pb[p + 1]), int(win_len[p]), int(halo_ptr[p]), int(halo_ptr[p + 1] - halo_ptr[p]))
            base = ps & ~1  # the LDS image starts at the even row at or below the partition start
            win = np.concatenate([x[base:ps + wl], x[halo_cols[hb:hb + hn]]])
            assert not (pb_assign and wl == 0 and hn == 0), "a partition without a window has a segment"
            for s in range(s0, s1):
                # record word 3: ELL pairs << 16 | inline residual pairs << 8 | column groups - 1
                npairs, ner, G = int(meta[s, 3] >> 16), int(meta[s, 3] >> 8) & 0xFF, int(meta[s, 3] & 0x3F) + 1
                p0, p1 = int(spp[s]), int(spp[s]) + npairs
                acc = np.zeros(64, dtype=np.float64)
                # the 16-byte record the kernel reads must agree with the prefix arrays
                assert meta[s, 0] == p0 and meta[s, 1] == scp[s] and meta[s, 2] == slab_row[s] and spp[s + 1] == p1 + ner
                assert (scp[s + 1] - scp[s]) == npairs * G + ner * 128
                assert ner == 0 or inline, "inline residual pairs only in the inline form"
                # which row every lane works on: plain = first row + lane; symmetric pairs = the
                # slab_lrow table (place in the partition's LDS image, 0xFFFF = none)
                if sym:
                    # every lane is covered exactly once by an adding lane of its own group:
                    # code 0/1/2 = adds for itself and the next 0/1/2 lanes, 3 = covered by a lane before
                    covered = np.zeros(64, dtype=int)
                    for l in np.flatnonzero(lane_code[s] != 3):
                        span = int(lane_code[s][l]) + 1
                        assert l + span <= 64 and np.all(lane_group[s][l:l + span] == lane_group[s][l])
                        covered[l:l + span] += 1
                    assert np.all(covered == 1), "group sums do not cover every lane exactly once"
                    lr = lrow_tab[s]
                    has = lr != 0xFFFF
                    assert np.all((lr[has] >= (ps & 1)) & (lr[has] < (ps & 1) + wl)), "lane row outside the partition"
                    rows_l = np.where(has, base + lr, -1)
                else:
                    r0 = int(slab_row[s])
                    has = np.arange(64) < pe - r0
                    rows_l = np.where(has, r0 + np.arange(64), -1)
                if p1 > p0:
                    v = ell_val[p0 * 128:p1 * 128].reshape(npairs, 64, 2)
                    words = ell_col[scp[s]:scp[s] + npairs * G].reshape(npairs, G)[:, lane_group[s]]  # [pair][lane]
                    c = np.stack([words & 0xFFFF, words >> 16], axis=2)
                    if meta[s, 3] & 0x80:
                        # the slab stores its columns relative to the lane's own place in the LDS image;
                        # lanes without a row read column 0
                        assert not sym
                        lrow_l = np.where(has, rows_l - base, 0)
                        c = np.where(has[None, :, None], (c + lrow_l[None, :, None]) & 0xFFFF, 0)
                    if sym:
                        # symmetric pair storage: bit 15 = "this entry also stands for its mirror image":
                        # value * x[row of the lane] goes to row `column` of the same partition
                        mirror = (c >> 15).astype(bool)
                        c = c & 0x7FFF
                        assert not mirror.any() or (c[mirror].max() < (ps & 1) + wl and c[mirror].min() >= (ps & 1)), "mirror target outside the partition's rows"
                        xrow = np.where(has, x[np.maximum(rows_l, 0)], 0.0)
                        contrib = v * xrow[None, :, None]
                        # lanes without a row read the last group's words: their values are zero
                        assert not (v[:, ~has, :]).any(), "a lane without a row holds a value"
                        np.add.at(y_mirror, base + c[mirror], contrib[mirror])
                    assert c.max() < len(win), "window-local column outside the window"
                    acc = (v * win[c]).sum(axis=(0, 2))
                y[rows_l[has]] = acc[has]
                written[rows_l[has]] += 1
                if ner:
                    # inline residual: values behind the ELL pairs, global columns [pair][2][lane]
                    # behind the shared column words; lanes without a row hold zeros
                    ve = ell_val[p1 * 128:(p1 + ner) * 128].reshape(ner, 64, 2)
                    ce = ell_col[scp[s] + npairs * G:scp[s + 1]].reshape(ner, 2, 64).transpose(0, 2, 1)
                    assert ce.max() < n and not np.any(ve[:, ~has, :])
                    y_inl[rows_l[has]] = (ve * x[ce]).sum(axis=(0, 2))[has]
    er_col = plan.array("er_col")
    er_val = plan.array("er_val")
    assert np.all(seg_done == 1), "every residual segment belongs to exactly one work item"
    if inline:
        # inline form (tiny residual): the pairs behind the slabs hold the same entries as the CSR segments
        y_seg = np.zeros(n)
        np.add.at(y_seg, np.repeat(seg_row & 0x7FFFFFFF, np.diff(seg_ptr)), er_val * x[er_col])
        assert np.allclose(y_inl, y_seg, rtol=0, atol=1e-12 * (np.abs(y_seg).max() + 1e-300))
    y += y_mirror
    if sym:
        # one workgroup holds a partition's accumulators: every item is exactly one partition
        assert np.all(items[:, 1] - items[:, 0] == 1)
        assert np.all(segs[:, 1] == np.searchsorted(slab_part, segs[:, 0])) and np.all(segs[:, 2] == np.searchsorted(slab_part, segs[:, 0], side="right"))
    y_csr = np.zeros(n)
    if len(seg_row):
        # (segments may be empty in the direct shape: one per row, y assigned)
        np.add.at(y_csr, np.repeat(seg_row & 0x7FFFFFFF, np.diff(seg_ptr)), er_val * x[er_col])
    if plan.stats["er_partials"] > 0:
        # panel form of the residual, walked as ehyb_pb_scale_kernel / ehyb_pb_reduce_kernel index it
        y_pb = walk_panel_residual(plan, x, written)
        tol = 1e-12 * (np.abs(er_val * x[er_col]).sum() / max(1, len(seg_row)) + np.abs(y_csr).max() + 1e-300)
        assert np.allclose(y_pb, y_csr, rtol=0, atol=tol), "panel form and CSR segments of the residual disagree"
        y += y_pb
    else:
        y += y_csr
    return y, written


def decode_panel_slots(colf, chunk, jump):
    """Slot of every entry from what pass 1 of the panel residual really streams (include/ehyb.h
    EHYB_ARR_PB_COLF / PB_CHUNK / PB_JUMP), computed as ehyb_pb_scale_kernel does per 64-entry chunk:
    slot = PB_JUMP[chunk's first jump + jumps up to the lane - 1] + pieces begun before the lane's (mod 2^32).
    -> (local column, slot) per entry; 0xFFFFFFFF = padding."""
    colf = colf.astype(np.int64).reshape(-1, 64)
    assert len(chunk) == len(colf) + 1 and chunk[-1] == len(jump)
    chunk = chunk.astype(np.int64)[:-1].reshape(-1, 1)
    jump = jump.astype(np.int64)
    head = (colf >> 15) & 1
    jmp = (colf >> 14) & 1
    assert np.all(head[:, 0] == 1) and np.all(jmp[:, 0] == 1) and np.all(jmp <= head)
    hcount = np.cumsum(head, axis=1)
    jcount = np.cumsum(jmp, axis=1)
    assert len(jump) == int(jmp.sum()) and (len(chunk) == 0 or chunk[0, 0] == 0) and np.all(chunk[1:, 0] == np.cumsum(jmp.sum(axis=1))[:-1])
    slot = (jump[chunk + jcount - 1] + hcount - 1) & 0xFFFFFFFF
    return (colf & 0x3FFF).reshape(-1), slot.reshape(-1)


def walk_panel_residual(plan, x, written=None):
    """y contribution of the residual in panel form (include/ehyb.h EHYB_ARR_PB_*): pass 1 per unit
    {first column, columns, first entry, end entry} multiplies with the unit's x panel and sums the
    entries of every slot; pass 2 per unit {first partial, end partial, first row, rows} adds the
    partials into the rows of its block.  Checks the invariants the kernels rely on."""
    n = plan.n
    val, col, dst = plan.array("pb_val"), plan.array("pb_col").astype(np.int64), plan.array("pb_dst").astype(np.int64)
    u1 = plan.array("pb_units1").reshape(-1, 4)
    u2 = plan.array("pb_units2").reshape(-1, 4)
    prow = plan.array("pb_row").astype(np.int64)
    P = plan.stats["er_partials"]
    assert len(val) == len(col) == len(dst) and len(val) % 64 == 0 and len(prow) == P
    # what the kernel streams (flags + chunk records + jump list) must decode to exactly these columns and slots
    col_k, dst_k = decode_panel_slots(plan.array("pb_colf"), plan.array("pb_chunk"), plan.array("pb_jump"))
    assert np.array_equal(col_k, col) and np.array_equal(dst_k, dst), "compressed slots of pass 1 do not decode to pb_dst"
    pad = dst == 0xFFFFFFFF
    assert not val[pad].any() and dst[~pad].max() < P and plan.stats["nnz_er"] == int((~pad).sum())
    # an entry's slot is shared only with neighbours inside its own 64-entry chunk (what the lane sums assume)
    live = np.flatnonzero(~pad)
    first = np.full(P, -1, dtype=np.int64)
    last = np.full(P, -1, dtype=np.int64)
    first[dst[live][::-1]] = live[::-1]
    last[dst[live]] = live
    assert first.min() >= 0, "a partial no entry writes"
    cnt = np.bincount(dst[live], minlength=P)
    assert np.all(last - first + 1 == cnt) and np.all(first // 64 == last // 64), "slot pieces must be contiguous inside a chunk"
    partial = np.zeros(P)
    covered = np.zeros(len(val), dtype=np.int32)
    for c0, nc, b, e in u1:
        assert b % 64 == 0 and e % 64 == 0 and e > b and 0 <= c0 and c0 + nc <= n
        win = x[c0:c0 + nc]
        k = np.arange(b, e)[~pad[b:e]]
        assert len(k) == 0 or col[k].max() < nc, "local column outside the staged panel"
        np.add.at(partial, dst[k], val[k] * win[col[k]])
        covered[b:e] += 1
    assert np.all(covered == 1), "every entry belongs to exactly one pass-1 unit"
    # one pass-1 workgroup = one item = a run of consecutive units; the items tile the unit list; with column segments
    # (multi-GPU) an item stays inside one segment
    it1 = plan.array("pb_items1").reshape(-1, 2)
    assert len(it1) > 0 and it1[0, 0] == 0 and it1[-1, 1] == len(u1) and np.all(it1[1:, 0] == it1[:-1, 1]) and np.all(it1[:, 1] > it1[:, 0])
    segf, segi = plan.array("col_seg_first"), plan.array("pb_seg_item")
    if len(segf) >= 2:
        assert len(segi) == len(segf) and segi[0] == 0 and segi[-1] == len(it1)
        for s in range(len(segf) - 1):
            if segi[s + 1] > segi[s]:
                uu = u1[it1[segi[s], 0]:it1[segi[s + 1] - 1, 1]]
                assert np.all(uu[:, 0] >= segf[s]) and np.all(uu[:, 0] + uu[:, 1] <= segf[s + 1]), "a pass-1 item straddles a column segment"
    y = np.zeros(n)
    seen = np.zeros(P, dtype=np.int32)
    rows_seen = np.zeros(n, dtype=np.int32)
    for pb, pe, r0, nr in u2:
        if nr < 0:
            # the block assigns: it is the only writer of its rows (their partitions have no window), also
            # where no partial arrives
            nr = -nr
            assert pe >= pb
            if written is not None:
                written[r0:r0 + nr] += 1
        else:
            assert pe > pb
        assert pe == pb or prow[pb:pe].max() < nr
        np.add.at(y, r0 + prow[pb:pe], partial[pb:pe])
        seen[pb:pe] += 1
        rows_seen[r0:r0 + nr] += 1
    assert np.all(seen == 1), "every partial belongs to exactly one pass-2 unit"
    assert rows_seen.max() <= 1, "row blocks of pass 2 must not overlap (their write-back is not atomic)"
    return y
