"""numpy restatement of the reference's EHYB *format* and of the way its kernels walk it.

TEST INFRASTRUCTURE ONLY (see oracle/ehyb_oracle.c for the rules and the parity status:
"parity unpinned" -- the reference has no fixtures and cannot be built in this image).

Restates, for small matrices:
  * reference_sizing      solver_test.c:158-182 (sym) / 53-77 (unsym) -- the original heuristic
  * num_in_row2           reordering.c:358-361
  * build_reference_ehyb  convert.c:61-146 (vecsGenBlockELL), 148-168 (vecsGenER),
                          170-311 (COO2EHYBCore), 316-369 (COO2EHYB)
  * walk_reference_ehyb   kernel.cu:150-163 (ELL slabs), 176-188 (ER slabs), 69-77 (vecReorderER)
`warp` is 32 in the reference (kernel.h:20); it is a parameter so the same rules can be
evaluated at the wave64 slab height of the MI355X build.

The reference's long-row path (rows with more than threadLongVec = 512 in-window entries,
convert.c:92-101) is broken; inputs that would take it are refused.
"""
import math

import numpy as np

THREAD_LONG_VEC = 512  # kernel.h:26
SM_SIZE, SM_SIZE2, THREAD_ELL, MAX_SHARED = 82, 80, 1024, 93 * 1024  # kernel.h:21-25


def reference_sizing(dimension, symmetric):
    """(nParts, vectorCacheSize, kernelPerPart) as solver_test.c:158-182 / 53-77 compute them.
    vectorCacheSize is int16 there: values past 32767 wrap."""
    def i16(v):
        v = int(v) & 0xFFFF
        return v - 0x10000 if v >= 0x8000 else v

    part_factor, kpp = 1, 1
    cache = i16(math.ceil(dimension / (part_factor * SM_SIZE * THREAD_ELL)) * THREAD_ELL)
    if cache < MAX_SHARED // (2 * 8):
        karr = [8, 5, 4, 2]
        k = 0
        kpp = karr[k]
        cache = i16(kpp * math.ceil(dimension / (SM_SIZE2 * THREAD_ELL)) * THREAD_ELL)
        k += 1
        while cache * 8 > MAX_SHARED and k < 4:
            kpp = karr[k]
            cache = i16(kpp * math.ceil(dimension / (SM_SIZE2 * THREAD_ELL)) * THREAD_ELL)
            k += 1
        nparts = (SM_SIZE2 if symmetric else SM_SIZE) // kpp
    else:
        while cache * 8 > MAX_SHARED:
            part_factor += 1
            cache = i16(math.ceil(dimension / (part_factor * SM_SIZE * THREAD_ELL)) * THREAD_ELL)
        nparts = part_factor * SM_SIZE
    return nparts, cache, kpp


def num_in_row2(row_idx, J, part_boundary, cache):
    """Entries of each row whose column lies in [partStart, partStart+cache) -- reordering.c:358-361."""
    n = len(row_idx) - 1
    out = np.zeros(n, dtype=np.int64)
    nparts = len(part_boundary) - 1
    for p in range(nparts):
        s, e = int(part_boundary[p]), int(part_boundary[p + 1])
        for r in range(s, e):
            cols = J[row_idx[r]:row_idx[r + 1]]
            out[r] = int(np.count_nonzero((cols >= s) & (cols < s + cache)))
    return out


def build_reference_ehyb(row_idx, J, V, part_boundary, cache, warp=32):
    """COO2EHYB (convert.c:316-369) on a permuted row-grouped matrix.  Returns a dict holding the
    arrays of matrixEHYB (spmv.h:35-63) plus the metrics the reference prints."""
    row_idx = np.asarray(row_idx, dtype=np.int64)
    J = np.asarray(J, dtype=np.int64)
    V = np.asarray(V, dtype=np.float64)
    n = len(row_idx) - 1
    nparts = len(part_boundary) - 1
    num = np.diff(row_idx)
    num2 = num_in_row2(row_idx, J, part_boundary, cache)
    if np.any(num2 > THREAD_LONG_VEC):
        raise ValueError("input would take the reference's (broken) long-row path")
    bpp = cache // warp  # blockPerPart, convert.c:80
    width = np.zeros(nparts * bpp, dtype=np.int64)
    num_er = np.zeros(n, dtype=np.int64)
    for p in range(nparts):  # vecsGenBlockELL, convert.c:87-135
        s, e = int(part_boundary[p]), int(part_boundary[p + 1])
        for it in range(bpp):
            b0 = s + it * warp
            rows = range(b0, min(b0 + warp, e))
            w = 0
            for r in rows:
                w = max(w, int(num2[r]))
                if num2[r] != num[r]:  # convert.c:115-119
                    num_er[r] = num[r] - num2[r]
            width[it + bpp * p] = w
        for r in range(s + bpp * warp, e):  # rows past the window: whole row to ER, convert.c:128-134
            num_er[r] += num[r]
    rows_er = int(np.count_nonzero(num_er))
    to_er = int(num_er.sum())
    bias = np.zeros(nparts * bpp, dtype=np.int64)  # convert.c:336-340
    bias[1:] = np.cumsum(warp * width)[:-1]
    size_ell = int((warp * width).sum())
    # vecsGenER, convert.c:148-168: all rows sorted by ER length, descending (qsort; ties made
    # deterministic here by a stable sort)
    order = np.argsort(-num_er, kind="stable")
    reorder_er = np.empty(n, dtype=np.int64)
    reorder_er[order] = np.arange(n)
    nblk_er = math.ceil(rows_er / warp) if rows_er else 0
    width_er = np.zeros(nblk_er, dtype=np.int64)
    row_vec_er = np.zeros(rows_er, dtype=np.int64)
    for r in range(n):
        if num_er[r] > 0:
            loc = int(reorder_er[r])
            row_vec_er[loc] = r
            width_er[loc // warp] = max(width_er[loc // warp], int(num_er[r]))
    bias_er = np.zeros(nblk_er, dtype=np.int64)  # convert.c:348-350
    if nblk_er:
        bias_er[1:] = np.cumsum(warp * width_er)[:-1]
    size_er = int((warp * width_er).sum())
    col_ell = np.zeros(size_ell, dtype=np.int64)
    val_ell = np.zeros(size_ell, dtype=np.float64)
    col_er = np.zeros(size_er, dtype=np.int64)
    val_er = np.zeros(size_er, dtype=np.float64)
    waste = 0
    for p in range(nparts):  # COO2EHYBCore, convert.c:207-308
        s, e = int(part_boundary[p]), int(part_boundary[p + 1])
        fetch_end = s + cache
        for it in range(bpp):
            blk = it + bpp * p
            w, b = int(width[blk]), int(bias[blk])
            for lane in range(warp):
                r = s + it * warp + lane
                k_ell = 0
                if r < e:
                    k_er = 0
                    if num_er[r] > 0:
                        loc = int(reorder_er[r])
                        b_er, lane_er = int(bias_er[loc // warp]), loc % warp
                    for k in range(int(row_idx[r]), int(row_idx[r + 1])):
                        if s <= J[k] < fetch_end:  # convert.c:247
                            col_ell[b + lane + k_ell * warp] = J[k] - s
                            val_ell[b + lane + k_ell * warp] = V[k]
                            k_ell += 1
                            assert k_ell <= w  # convert.c:251-254
                        else:
                            assert num_er[r] > 0  # convert.c:256-259
                            col_er[b_er + lane_er + k_er * warp] = J[k]
                            val_er[b_er + lane_er + k_er * warp] = V[k]
                            k_er += 1
                waste += w - k_ell  # zero fill, convert.c:269-281
        for r in range(s + bpp * warp, e):  # convert.c:285-305
            loc = int(reorder_er[r])
            b_er, lane_er = int(bias_er[loc // warp]), loc % warp
            for q, k in enumerate(range(int(row_idx[r]), int(row_idx[r + 1]))):
                col_er[b_er + lane_er + q * warp] = J[k]
                val_er[b_er + lane_er + q * warp] = V[k]
    return dict(n=n, nparts=nparts, cache=cache, warp=warp, part_boundary=np.asarray(part_boundary, dtype=np.int64),
                width=width, bias=bias, col_ell=col_ell, val_ell=val_ell, rows_er=rows_er, row_vec_er=row_vec_er,
                width_er=width_er, bias_er=bias_er, col_er=col_er, val_er=val_er, to_er=to_er,
                nnz_ell=int(num.sum()) - to_er, waste=waste, size_block_ell=size_ell, size_er=size_er)


def walk_reference_ehyb(L, x):
    """What one launch of kernelCachedBlockedELL + vecReorderER computes, on the CPU."""
    n, warp, cache = L["n"], L["warp"], L["cache"]
    bpp = cache // warp
    x = np.asarray(x, dtype=np.float64)
    y = np.zeros(n, dtype=np.float64)
    pb = L["part_boundary"]
    for p in range(L["nparts"]):
        s, e = int(pb[p]), int(pb[p + 1])
        cached = np.zeros(cache, dtype=np.float64)  # kernel.cu:137-138 (clipped to n: the
        m = min(cache, n - s)                       # reference reads past the end, a-10 item 5)
        cached[:m] = x[s:s + m]
        for it in range(bpp):  # kernel.cu:150-163
            blk = it + bpp * p
            w, b = int(L["width"][blk]), int(L["bias"][blk])
            for lane in range(warp):
                r = s + it * warp + lane
                if r < e:
                    dot = 0.0
                    for k in range(w):
                        idx = b + warp * k + lane
                        dot += L["val_ell"][idx] * cached[L["col_ell"][idx]]
                    y[r] = dot
    y_er = np.zeros(L["rows_er"], dtype=np.float64)
    for blk in range(len(L["width_er"])):  # kernel.cu:176-188
        w, b = int(L["width_er"][blk]), int(L["bias_er"][blk])
        for lane in range(warp):
            r = blk * warp + lane
            if r < L["rows_er"]:
                dot = 0.0
                for k in range(w):
                    idx = b + lane + warp * k
                    dot += L["val_er"][idx] * x[L["col_er"][idx]]
                y_er[r] = dot
    for i in range(L["rows_er"]):  # kernel.cu:69-77
        y[L["row_vec_er"][i]] += y_er[i]
    return y
