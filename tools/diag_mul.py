#!/usr/bin/env python3
"""Which layers of a bf16 train step multiply their incoming gradient by the saved activation derivative in a
launch of its own (ops.mul_bf16) instead of having the upstream data-gradient kernel apply it:
python tools/diag_mul.py [--stage 2] [--batch 4]"""
import argparse, os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import music_generator_amd as M
from music_generator_amd import ops, layers, backend as K

ap = argparse.ArgumentParser()
ap.add_argument("--stage", type=int, default=2)
ap.add_argument("--batch", type=int, default=4)
args = ap.parse_args()
K.set_precision("bfloat16")
M.clear_session()
d = M.define_discriminator(args.stage + 1)
g = M.define_generator(args.stage + 1)
gan = M.define_composite(d, g, (1, 100, 2))[args.stage][1]
gan.set_train_steps(100)
h, w = 4 << args.stage, 750 << args.stage
real = torch.rand((args.batch, h, w, 2), device="cuda") * 2 - 1
seen = collections.Counter()
orig = layers._Contraction._backward_bf16


def spy(self, saved, dy, need, existing, wgrad, outmul):
    if saved.dact is not None and not saved.premultiplied:
        seen[(self.name, tuple(saved.x.shape), self.filters, self.kernel_size, type(dy).__name__, str(dy.dtype))] += 1
    return orig(self, saved, dy, need, existing, wgrad, outmul)


layers._Contraction._backward_bf16 = spy
gan.train_step(real)
torch.cuda.synchronize()
for k, v in sorted(seen.items()):
    print(v, k)
