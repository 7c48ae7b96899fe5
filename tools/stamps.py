#!/usr/bin/env python3
"""Diagnostic: per-workgroup timeline of the ELL kernel (stamped instantiation) and the
streaming-read ceiling of the device.  Not part of the product path.

usage: python tools/stamps.py [--workload audikw_1-like] [--lds 10240] [--threads 1024] [--items 2]
"""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="audikw_1-like")
    ap.add_argument("--lds", type=int, default=0, help="window budget in doubles (0 = the mode's default)")
    ap.add_argument("--sym", type=int, default=0, help="1 = symmetric pair storage (one workgroup per partition)")
    ap.add_argument("--part-rows", type=int, default=0)
    ap.add_argument("--threads", type=int, default=1024)
    ap.add_argument("--items", type=int, default=2)
    ap.add_argument("--variant", type=int, default=0, help="0/1 LDS slab counter, 3 static round-robin")
    ap.add_argument("--xcd-map", type=int, default=1, help="1 = every XCD takes one contiguous run of items (default), 2 = blockIdx order")
    ap.add_argument("--dump", default="", help="write every item's duration, slab range and placement to this .npz (offline analysis)")
    ap.add_argument("--graph-compress", type=int, default=0, help="cfg.graph_compress (A/B of the partitions)")
    ap.add_argument("--triple-gather", type=int, default=0, help="1 = the stamped launch stages every window entry from THREE vectors (what folding CG's direction "
                                                                 "update into the staging would gather): the price of that fold")
    args = ap.parse_args()
    import numpy as np

    import bench as B
    import ehyb_spmv_gpu_amd as E
    from ehyb_spmv_gpu_amd import _lib

    lib = _lib.load()
    bw = C.c_double()
    for mb in (256, 1024, 4096):
        lib.ehyb_measure_read_bw(mb << 20, 20, C.byref(bw))
        print(f"streaming read of {mb} MiB: {bw.value:.0f} GB/s")

    gen, gargs, _ = B.WORKLOADS[args.workload]
    kw = dict(lds_doubles=args.lds, threads=args.threads, items_per_cu=args.items, ell_variant=args.variant, sym_pairs=args.sym, xcd_map=args.xcd_map, graph_compress=args.graph_compress)
    if args.part_rows:
        kw["part_rows"] = args.part_rows
    cfg = E.make_config(**kw)
    m = E.Matrix.generate(gen, *gargs, cfg=cfg)
    x = E.x_glibc(m.n)
    m.reorder(cfg)
    xp = E.vector_reorder(x, m.reorder_list)
    plan = E.Plan(m, cfg)
    st = plan.stats
    dx, dy = E.DeviceBuffer(m.n).upload(xp), E.DeviceBuffer(m.n)
    for _ in range(5):
        plan.spmv(dx.ptr, dy.ptr)
    n_items = st["n_items"]
    out = np.zeros(n_items * 4, dtype=np.uint64)
    fn = lib.ehyb_debug_ell_stamps_probe
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    for rep in range(3):
        rc = fn(plan.h, C.c_void_p(dx.ptr), C.c_void_p(dy.ptr), out.ctypes.data_as(C.c_void_p), args.triple_gather)
        assert rc == 0, lib.ehyb_last_error()
    s = out.reshape(-1, 4).astype(np.int64)
    t0 = s[:, 0].min()
    start, staged, end, xcc = (s[:, 0] - t0) / 100.0, (s[:, 1] - t0) / 100.0, (s[:, 2] - t0) / 100.0, s[:, 3]
    items = plan.array("items").reshape(-1, 8)
    # plain storage: workgroup b took item xcd_item(b) (ehyb_hip.hip); symmetric pairs: item b (heaviest first)
    if args.xcd_map != 2 and st["sym_pairs"] == 0:
        b = np.arange(n_items)
        k, j, chunk, rem = b & 7, b >> 3, n_items >> 3, n_items & 7
        items = items[k * chunk + np.minimum(k, rem) + j]
    spp = plan.array("slab_pair_ptr").astype(np.int64)
    pairs = spp[items[:, 3]] - spp[items[:, 2]]
    segs = plan.array("segs").reshape(-1, 8)
    first = segs[items[:, 0]]  # first segment of every item: {part, slab range, halo count, row range, window, halo base}
    rows, halo = (first[:, 5] - first[:, 4]).astype(np.int64), first[:, 3].astype(np.int64)
    dur = end - start
    A = np.stack([pairs, rows, halo, np.ones(n_items)], axis=1).astype(np.float64)
    coef, *_ = np.linalg.lstsq(A, dur, rcond=None)
    fit = A @ coef
    print(f"duration ~ {coef[0] * 1e3:.2f} ns/pair-row + {coef[1] * 1e3:.2f} ns/row + {coef[2] * 1e3:.2f} ns/halo column + {coef[3]:.1f} us; "
          f"residual rms {np.sqrt(np.mean((dur - fit) ** 2)):.2f} us; corr(duration, pairs) {np.corrcoef(dur, pairs)[0, 1]:.2f} "
          f"rows {np.corrcoef(dur, rows)[0, 1]:.2f} halo {np.corrcoef(dur, halo)[0, 1]:.2f}")
    print(f"items {n_items}  kernel span {end.max():.1f} us")
    print(f"start   : min {start.min():.1f} med {np.median(start):.1f} max {start.max():.1f} us")
    print(f"staging : min {(staged - start).min():.1f} med {np.median(staged - start):.1f} max {(staged - start).max():.1f} us")
    print(f"compute : min {(end - staged).min():.1f} med {np.median(end - staged):.1f} max {(end - staged).max():.1f} us")
    print(f"end     : min {end.min():.1f} p10 {np.percentile(end, 10):.1f} med {np.median(end):.1f} p90 {np.percentile(end, 90):.1f} max {end.max():.1f} us")
    rate = pairs * 64 * 20 / ((end - staged) * 1e-6) / 1e9
    print(f"per-WG stream rate: min {rate.min():.1f} med {np.median(rate):.1f} max {rate.max():.1f} GB/s; pairs/item min {pairs.min()} max {pairs.max()}")
    for xc in range(8):
        sel = xcc == xc
        if sel.any():
            print(f"  xcc {xc}: {sel.sum():4d} WGs, end med {np.median(end[sel]):.1f} max {end[sel].max():.1f} us, bytes {pairs[sel].sum() * 64 * 20 / 1e6:.1f} MB")
    if args.dump:
        np.savez(args.dump, dur=dur, start=start, staged=staged, end=end, xcc=xcc, slab_begin=items[:, 2], slab_end=items[:, 3], pairs=pairs, rows=rows, halo=halo)
    meta3 = plan.array("slab_meta").reshape(-1, 4)[:, 3].astype(np.int64)
    for i in np.argsort(-dur)[:8]:
        w2 = meta3[items[i, 2]:items[i, 3]] >> 16
        g = (meta3[items[i, 2]:items[i, 3]] & 0x3F) + 1
        print(f"  slow item {i}: duration {dur[i]:.1f} us (staging {staged[i] - start[i]:.1f}) segments {items[i, 1] - items[i, 0]} slabs {items[i, 3] - items[i, 2]} pairs {pairs[i]} "
              f"widest slab {w2.max()} pairs, mean groups {g.mean():.1f}, rows {rows[i]} halo {halo[i]} xcc {xcc[i]}")
    late = np.argsort(-end)[:8]
    for i in late:
        print(f"  late item {i}: segments {items[i, 1] - items[i, 0]} slabs {items[i, 3] - items[i, 2]} pairs {pairs[i]} start {start[i]:.1f} staged {staged[i]:.1f} end {end[i]:.1f} xcc {xcc[i]}")


if __name__ == "__main__":
    main()
