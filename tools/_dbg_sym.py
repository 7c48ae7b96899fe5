import sys, numpy as np
sys.path.insert(0, "/root/repo")
import ehyb_spmv_gpu_amd as E
kw = dict()
cfg = E.make_config(**kw)
m = E.Matrix.generate("rmat", 18, 1 << 22, 1, cfg=cfg)
m.reorder(cfg)
host = E.Plan(m, E.make_config(symbolic=1, **kw))
dev = E.Plan(m, E.make_config(symbolic=2, **kw))
for name in ["pb_val", "pb_col", "pb_dst", "pb_colf", "pb_chunk", "pb_jump", "pb_row", "pb_units1", "pb_items1", "pb_units2", "pb_src"]:
    a, b = host.array(name), dev.array(name)
    if a.shape != b.shape:
        print(name, "shape", a.shape, b.shape); continue
    d = np.nonzero(a != b)[0]
    print(name, len(a), "mismatches", len(d), "first", d[:5], a[d[:5]], b[d[:5]])
J = m.J; rp = m.row_idx
asc = 0
for r in range(0, m.n, 997):
    seg = J[rp[r]:rp[r+1]]
    asc += int(np.any(np.diff(seg) < 0))
print("sampled rows not ascending:", asc)
