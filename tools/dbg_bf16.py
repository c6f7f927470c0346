import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import music_generator_amd as M
from oracle import ref_step as R
from test_gpu_step import build
stage, twin = 1, 0
d, g, comp, W = build(stage + 1)
model = g[stage][twin]
_, z4 = R.synthetic_inputs(stage, 2, seed=0)
acts = {}
for prec in ("float32", "bfloat16"):
    M.backend.set_precision(prec)
    y, tape = model.forward(torch.from_numpy(z4[0]).cuda(), save=True)
    ent = [t for t in tape if isinstance(t, tuple) and len(t) == 4 and isinstance(t[2], tuple)][0]
    acts[prec] = ent[0].float().cpu().numpy().reshape(-1, 4104, 32)
a, b = acts["bfloat16"], acts["float32"]
print(a.shape, b.shape)
segs = [("X", 0, 2), ("F", 8, 136)] + [("br%d" % k, 136 + 352 * k, 136 + 352 * (k + 1)) for k in range(2)] + [("O1", 136, 168), ("O2", 168, 232), ("O3", 232, 488), ("S", 2248, 2760), ("T1", 2760, 3784), ("T2", 3784, 4040), ("T3", 4040, 4072)]
for name, r0, r1 in segs:
    da, db = a[:, r0:r1], b[:, r0:r1]
    print("%-4s rel %.3e  norms %.3e %.3e" % (name, np.linalg.norm(da - db) / (np.linalg.norm(db) + 1e-30), np.linalg.norm(da), np.linalg.norm(db)))
# per-row check in T3 of tile 0
for row in range(4040, 4048):
    print(row, a[0, row, :4], b[0, row, :4])
