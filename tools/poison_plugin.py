"""pytest plugin: the LDS of every CU is filled with NaN words in front of every library call, so that a kernel
that reads LDS it did not write itself shows up as NaN / changed results in the parity tests:
    hipcc -O3 --offload-arch=gfx950 -shared -fPIC tools/poison/lds_poison.hip -o tools/poison/liblds_poison.so
    PYTHONPATH=tools:. python -m pytest -p poison_plugin tests -m gpu -q
"""
import ctypes
import os

import torch


def pytest_configure(config):
    from music_generator_amd import ops
    lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "poison", "liblds_poison.so"))
    lib.lds_poison.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    lib.lds_poison.restype = ctypes.c_int
    sink = torch.zeros((4,), dtype=torch.int32, device="cuda")
    real = ops._lib

    def poisoned():
        rc = lib.lds_poison(torch.cuda.current_stream().cuda_stream, sink.data_ptr())
        if rc:
            raise RuntimeError("lds_poison failed (%d)" % rc)
        return real()

    ops._lib = poisoned
    config._lds_sink = sink
