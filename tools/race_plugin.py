"""pytest plugin for a robustness check of the multi-stream / graph-replayed step: convolution, weight-gradient
and blend launches are preceded (at random) by a spin kernel of random length on their stream, so that the relative timing of the lanes
changes from launch to launch.  The bit-identity tests of tests/test_gpu_step.py must still pass:
    PYTHONPATH=tools:. python -m pytest -p race_plugin tests/test_gpu_step.py -m gpu -q -k "stream or graph or merged"
"""
import random

import torch


def pytest_configure(config):
    from music_generator_amd import ops
    rng = random.Random(1234)

    def wrap(name, prob, longest):
        orig = getattr(ops, name)

        def delayed(*a, **k):
            if rng.random() < prob:
                torch.cuda._sleep(rng.randrange(1000, longest))
            return orig(*a, **k)

        setattr(ops, name, delayed)

    wrap("conv_gemm", 0.4, 2000000)          # up to ~1 ms
    wrap("conv_wgrad", 0.4, 2000000)
    wrap("conv_gemm_bf16", 0.4, 2000000)
    wrap("weighted_sum_fwd", 0.5, 500000)
