#!/usr/bin/env python3
"""Exchange-volume and timeline model of the strong-scaling step (BASELINE config 5: ONE R-MAT sharded over N GPUs), no GPU needed.

For N = 2 / 4 / 8 row blocks of equal cost (the cuts `bench.py --gpus N` uses: ehyb_gen_rmat_block) it looks at every off-diagonal
block A[r, s] -- the entries of rank r's rows in rank s's columns -- and prices four ways of getting its product into y_r:

  row        (what dist.py does)  r multiplies; s sends the DISTINCT COLUMNS of the block:         nzc(r, s) doubles  s -> r
  column                          s multiplies with its own x; sends one partial sum per DISTINCT ROW:  nzr(r, s) doubles  s -> r
  two-sided                       per block whichever is smaller:                                   min(nzc, nzr)
  cover                           per block a vertex cover of its bipartite graph: the hub columns travel as x, the rows that still
                                  have an uncovered entry travel as partial sums (greedy by degree; the optimum is a minimum vertex
                                  cover, Koenig) -- the least any scheme can move for a 1-D distribution of x and y
  hub-replicated (1.5-D)          the H hottest columns of the WHOLE matrix are broadcast to every rank in one collective, the rest
                                  as `row`

Either way the data of block (r, s) flows s -> r over the one xGMI link of that pair, so the arms differ in volume and in WHEN the
sender can start: x entries are ready at the start of the step, partial sums when the sender has multiplied the block.

Timeline (assumptions stated in the output, not measurements): local rate, per-link rate and per-collective latency as arguments.

usage: python tools/dist_volume_model.py [--scale 22] [--ranks 2,4,8] [--hub 65536] [--out profiles/r04_dist_volume_rmatNN.json]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def block_stats(I, J, cuts):
    """Per (r, s): entries, distinct columns, distinct rows -- three N x N arrays."""
    N = len(cuts) - 1
    br = (np.searchsorted(cuts, I, side="right") - 1).astype(np.int64)
    bs = (np.searchsorted(cuts, J, side="right") - 1).astype(np.int64)
    blk = br * N + bs
    nnz = np.bincount(blk, minlength=N * N).reshape(N, N)
    n = int(cuts[-1])
    # distinct columns per block: unique (r, column) pairs (the column fixes s); distinct rows: unique (row, s) pairs
    ucol = np.unique(br * n + J)
    nzc = np.bincount((ucol // n) * N + (np.searchsorted(cuts, ucol % n, side="right") - 1), minlength=N * N).reshape(N, N)
    urow = np.unique(I.astype(np.int64) * N + bs)
    nzr = np.bincount((np.searchsorted(cuts, urow // N, side="right") - 1) * N + urow % N, minlength=N * N).reshape(N, N)
    return nnz, nzc, nzr, br, bs


def greedy_cover(I, J, sel):
    """Size of a vertex cover of the bipartite graph of the entries `sel` (rows on one side, columns on the other): columns whose
    degree in the block is above a threshold travel as x, every row that still has an entry in another column travels as a partial
    sum; the best of a handful of thresholds.  -> (cover size, columns in it, rows in it, entries multiplied by the RECEIVER)"""
    i, j = I[sel], J[sel]
    if len(i) == 0:
        return 0, 0, 0, 0
    cu, cinv, cdeg = np.unique(j, return_inverse=True, return_counts=True)
    best = None
    for t in (1, 2, 3, 4, 6, 8, 12, 16, 32, 64, 1 << 30):
        hot = cdeg[cinv] >= t                      # entry sits in a column that travels
        ncol = int((cdeg >= t).sum())
        nrow = len(np.unique(i[~hot]))
        size = ncol + nrow
        if best is None or size < best[0]:
            best = (size, ncol, nrow, int(hot.sum()))
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=int, default=22)
    ap.add_argument("--ranks", default="2,4,8")
    ap.add_argument("--hub", type=int, default=0, help="columns replicated on every rank in the 1.5-D arm (0: n / 256)")
    ap.add_argument("--local-TBps", type=float, default=4.6, help="rate at which a rank's plan streams its format bytes (single-GPU panel form)")
    ap.add_argument("--link-GBps", type=float, default=50.0, help="achieved rate of one xGMI link, one direction")
    ap.add_argument("--latency-us", type=float, default=15.0, help="start-up of one exchange (collective launch to first byte)")
    ap.add_argument("--single-us", type=float, default=0.0, help="measured single-GPU time of the whole matrix (0: format bytes of one plan / local rate)")
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    import ehyb_spmv_gpu_amd as E

    cfg = E.make_config(partitioner=E.EHYB_PART_DEGREE)
    t0 = time.time()
    m = E.Matrix.generate("rmat", args.scale, 1 << (args.scale + 3), 1, cfg=cfg)
    n, nnz = m.n, m.nnz
    I, J = m.I.astype(np.int64), m.J.astype(np.int64)
    print(f"# R-MAT 2^{args.scale}: {n} rows, {nnz} entries ({time.time() - t0:.1f}s)", file=sys.stderr)
    # cost model of a rank's plan (read off the host plans of round 3): 18 B per entry + 34 B per row
    B_ENTRY, B_ROW = 18.0, 34.0
    single_us = args.single_us or (B_ENTRY * nnz * 0.93 + B_ROW * n) / (args.local_TBps * 1e6)   # one plan packs its partials denser (2.5 vs 3.2 GB at 2^24)
    col_deg = np.bincount(J, minlength=n)
    H = args.hub or max(1024, n // 256)
    hub_cols = np.argsort(-col_deg, kind="stable")[:H]
    is_hub = np.zeros(n, dtype=bool)
    is_hub[hub_cols] = True
    out = {"matrix": f"rmat-{args.scale}", "rows": n, "nnz": nnz, "assumptions": {"local_TBps": args.local_TBps, "link_GBps_one_direction": args.link_GBps,
           "latency_us_per_exchange": args.latency_us, "bytes_per_entry": B_ENTRY, "bytes_per_row": B_ROW, "single_gpu_us": round(single_us, 1),
           "note": "a model: rates and latency are arguments, not measurements (no multi-GPU box)"}, "hub_columns": H,
           "hub_share_of_entries": round(float(col_deg[hub_cols].sum()) / nnz, 4), "by_ranks": {}}
    for N in [int(v) for v in args.ranks.split(",")]:
        mb = E.Matrix.generate("rmat_block", args.scale, 1 << (args.scale + 3), 1, 0, N, cfg=cfg)
        cuts = np.asarray(mb.block_cuts, dtype=np.int64)
        mb.free()
        t0 = time.time()
        bn, nzc, nzr, br, bs = block_stats(I, J, cuts)
        off = ~np.eye(N, dtype=bool)
        rows_of = np.diff(cuts)
        arms = {}
        # ---- volumes (doubles received per rank, and over the busiest link)
        vol = {"row": nzc * off, "column": nzr * off, "two-sided": np.minimum(nzc, nzr) * off}
        cover = np.zeros((N, N), dtype=np.int64)
        cover_recv_entries = np.zeros((N, N), dtype=np.int64)     # entries of the block the RECEIVER multiplies (their columns travel)
        blk = br * N + bs
        order = np.argsort(blk, kind="stable")
        first = np.concatenate(([0], np.cumsum(np.bincount(blk, minlength=N * N))))
        for r in range(N):
            for s in range(N):
                if r == s:
                    continue
                sel = order[first[r * N + s]:first[r * N + s + 1]]
                size, ncol, nrow, ent = greedy_cover(I, J, sel)
                cover[r, s] = size
                cover_recv_entries[r, s] = ent
        vol["cover"] = cover
        # 1.5-D: hub columns to everyone (each rank receives the H - own hubs it does not own), the other columns as `row`
        hub_owner = np.bincount(np.searchsorted(cuts, hub_cols, side="right") - 1, minlength=N)
        nonhub = ~is_hub[J]
        _, nzc_nh, _, _, _ = block_stats(I[nonhub], J[nonhub], cuts)
        vol["hub-replicated"] = nzc_nh * off + np.tile(hub_owner, (N, 1)) * off
        for name, v in vol.items():
            recv = v.sum(axis=1)
            arms[name] = {"doubles_received_per_rank": [int(x) for x in recv], "MB_received_max": round(float(recv.max()) * 8 / 1e6, 2),
                          "MB_over_the_busiest_link": round(float(v.max()) * 8 / 1e6, 2), "MB_total": round(float(v.sum()) * 8 / 1e6, 2)}
        # ---- timelines, microseconds
        link = args.link_GBps * 1e3      # bytes per microsecond
        lat = args.latency_us
        rate = args.local_TBps * 1e6     # bytes per microsecond
        ent_row = bn.sum(axis=1)         # entries of a rank's rows
        t_plan = (B_ENTRY * ent_row + B_ROW * rows_of) / rate
        own = np.diag(bn)
        pass2 = 0.12 * t_plan            # the closing pass over the rank's partial sums (12 of ~100 us at 8 ranks, round 3)

        def wire(v, r):                  # time until everything rank r receives in one exchange has landed: its busiest incoming link
            return lat + float(v[r].max()) * 8 / link if v[r].max() > 0 else 0.0

        def t_row_chunked(shares=(0.25, 0.75), hot_work=0.75):
            """dist.py's step: own columns | hot chunk | cold chunk | pass 2, the chunks back to back on the wire"""
            ends = []
            for r in range(N):
                ghost_t = t_plan[r] * (1 - own[r] / max(1, ent_row[r])) - pass2[r]
                own_t = t_plan[r] * own[r] / max(1, ent_row[r])
                w0 = lat + float((vol["row"][r] * shares[0]).max()) * 8 / link
                w1 = w0 + lat * 0.3 + float((vol["row"][r] * shares[1]).max()) * 8 / link     # the second group is already enqueued
                t = 3.0 + max(own_t, w0)
                t = max(t + ghost_t * hot_work, 3.0 + w1) + ghost_t * (1 - hot_work)
                ends.append(t + pass2[r])
            return max(ends)

        def t_two_sided(v, recv_entries):
            """per block the cheaper direction: a rank first multiplies what needs only its own x (its diagonal block and the blocks it
            computes FOR others, whose partial sums leave as soon as they exist), then the blocks whose columns have arrived meanwhile,
            then adds the partial sums that arrived; recv_entries[r, s] = entries of block (r, s) that r multiplies itself"""
            send_entries = bn - recv_entries                                   # entries of block (r, s) multiplied by s
            work_first = np.array([own[s] + (send_entries[:, s] * off[:, s]).sum() for s in range(N)], dtype=np.float64)
            work_second = np.array([(recv_entries[r] * off[r]).sum() for r in range(N)], dtype=np.float64)
            foreign_rows = np.array([(np.minimum(nzr, v)[:, s] * off[:, s]).sum() for s in range(N)], dtype=np.float64)
            t_first = (B_ENTRY * work_first + B_ROW * (rows_of + foreign_rows)) / rate * 0.88
            t_second = B_ENTRY * work_second / rate * 0.88
            ends = []
            for r in range(N):
                x_here = 3.0 + wire(v, r)                                      # x entries leave at once
                y_here = max(3.0 + t_first[s] for s in range(N) if s != r) + wire(v, r) * 0.5 if N > 1 else 0.0   # partial sums leave when their sender is done
                t = max(3.0 + t_first[r], x_here) + t_second[r]
                ends.append(max(t, y_here) + pass2[r] + 4.0)                   # + the add of the received partial sums
            load = t_first + t_second
            return max(ends), float(load.max() / load.mean())

        recv_row_all = bn * off
        recv_two = np.where(nzc <= nzr, bn, 0) * off
        tl = {"row, one exchange then everything (round 2)": max(3.0 + wire(vol["row"], r) + t_plan[r] for r in range(N)),
              "row, two chunks 0.25 / 0.75 pipelined (round 3, the default)": t_row_chunked()}
        t, lb = t_two_sided(vol["two-sided"], recv_two)
        tl["two-sided (row or column per block)"] = t
        arms["two-sided"]["compute_imbalance_max_over_mean"] = round(lb, 3)
        t, lb = t_two_sided(vol["cover"], cover_recv_entries)
        tl["cover (hub columns as x, the rest as partial sums, per block)"] = t
        arms["cover"]["compute_imbalance_max_over_mean"] = round(lb, 3)
        # the same cover with the rows cut for ITS cost (ehyb_gen_rmat_block_cost, cost model 1: an entry counts for the owner of its row
        # if its column has the higher degree of the two, else for the owner of its column) -- what bench.py does for --exchange cover
        mb1 = E.Matrix.generate("rmat_block", args.scale, 1 << (args.scale + 3), 1, 0, N, 1, cfg=cfg)
        cuts1 = np.asarray(mb1.block_cuts, dtype=np.int64)
        mb1.free()
        saved = (bn, nzc, nzr, off, rows_of, own, t_plan, pass2, ent_row)
        bn, nzc, nzr, br1, bs1 = block_stats(I, J, cuts1)
        rows_of = np.diff(cuts1)
        ent_row = bn.sum(axis=1)
        own = np.diag(bn)
        t_plan = (B_ENTRY * ent_row + B_ROW * rows_of) / rate
        pass2 = 0.12 * t_plan
        blk1 = br1 * N + bs1
        order1 = np.argsort(blk1, kind="stable")
        first1 = np.concatenate(([0], np.cumsum(np.bincount(blk1, minlength=N * N))))
        cover1 = np.zeros((N, N), dtype=np.int64)
        cre1 = np.zeros((N, N), dtype=np.int64)
        for r in range(N):
            for s in range(N):
                if r != s:
                    size, ncol, nrow, ent = greedy_cover(I, J, order1[first1[r * N + s]:first1[r * N + s + 1]])
                    cover1[r, s], cre1[r, s] = size, ent
        t, lb = t_two_sided(cover1, cre1)
        tl["cover, rows cut for the cover's own cost (what bench.py runs)"] = t
        recv1 = cover1.sum(axis=1)
        arms["cover, re-cut"] = {"rows_per_rank": [int(x) for x in rows_of], "doubles_received_per_rank": [int(x) for x in recv1], "MB_received_max": round(float(recv1.max()) * 8 / 1e6, 2),
                                 "MB_over_the_busiest_link": round(float(cover1.max()) * 8 / 1e6, 2), "MB_total": round(float(cover1.sum()) * 8 / 1e6, 2),
                                 "compute_imbalance_max_over_mean": round(lb, 3)}
        bn, nzc, nzr, off, rows_of, own, t_plan, pass2, ent_row = saved
        tl["no exchange at all (the plans alone: the ceiling of any 1-D scheme)"] = float(t_plan.max())
        res = {"rows_per_rank": [int(x) for x in rows_of], "entries_per_rank": [int(x) for x in ent_row], "own_block_share": [round(float(own[r]) / max(1, ent_row[r]), 3) for r in range(N)],
               "volume": arms, "timeline_us": {k: round(v, 1) for k, v in tl.items()}, "speedup_vs_one_gpu": {k: round(single_us / v, 2) for k, v in tl.items()},
               "local_plan_us_per_rank": [round(float(x), 1) for x in t_plan], "seconds": round(time.time() - t0, 1)}
        out["by_ranks"][str(N)] = res
        print(f"# N = {N}: " + "; ".join(f"{k.split(' (')[0]} {v['MB_received_max']} MB" for k, v in arms.items()), file=sys.stderr)
        for k, v in tl.items():
            print(f"#     {v:8.1f} us  {single_us / v:5.2f} x   {k}", file=sys.stderr)
    m.free()
    txt = json.dumps(out, indent=1)
    if args.out:
        open(args.out, "w").write(txt + "\n")
    print(txt)


if __name__ == "__main__":
    main()
