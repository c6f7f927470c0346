import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import music_generator_amd as M
from oracle import ref_step as R, bf16_emulation as E
from test_gpu_step import build
from test_gpu_bf16 import _blocks_model
def bf(a): return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(torch.bfloat16).float().numpy()
stage = 1
d, g, comp, W = build(stage + 1)
model = _blocks_model(d[stage][0])
real, _ = R.synthetic_inputs(stage, 4, seed=0)
names = R.layer_names(stage + 1)
Wt = {k: torch.tensor(np.asarray(v), dtype=torch.float64) for k, v in W.items()}
blk = names["d"][stage]; _, soft = R.activation_alphas(stage)
seq = [(blk["from_audio"], {})] + [(c, ({"strides": (2, 2), "padding": "valid"} if i == 8 else {})) for i, c in enumerate(blk["convs"])] + [(blk["to128"], {})]
M.backend.set_precision("bfloat16")
x = torch.from_numpy(real).cuda()
xo = torch.tensor(real, dtype=torch.float64)
for op, (name, kw) in zip(model._plan, seq):
    assert op.node.layer.name == name, (op.node.layer.name, name)
    x, _ = op.node.layer.forward([x], False, act_alpha=op.act_node.layer.alpha_value)
    xin = xo
    xo = E._conv_sr(Wt, name, xo, soft, **kw)
    # the same layer applied by the oracle to the DEVICE's input: isolates this layer
    dv = x.float().cpu().numpy(); ref = xo.numpy()
    print(name, tuple(dv.shape), str(x.dtype)[6:], "alpha", op.act_node.layer.alpha_value, soft,
          "chain rel-L2 %.2e" % (np.linalg.norm(dv - ref) / np.linalg.norm(ref)), "mismatch frac %.2e" % (dv != bf(ref)).mean(),
          "rms|ref| %.3f" % np.sqrt((ref**2).mean()))
