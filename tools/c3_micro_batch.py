"""The full-resolution bf16 step of bench.py (c3_run) with another micro-batch size: speed and peak memory.
PYTHONPATH=. python tools/c3_micro_batch.py <batch> <micro batch>"""
import json
import sys

import torch

import bench
import music_generator_amd as M
from music_generator_amd import ops

torch.cuda.set_device(0)
r = bench.c3_run(torch, M, ops, int(sys.argv[1]), int(sys.argv[2]))
print(json.dumps({k: r[k] for k in ("workload", "value", "ms_per_step", "peak_hbm_gib", "step_tflops_algorithmic")}))
