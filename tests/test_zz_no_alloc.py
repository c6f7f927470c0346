"""mgx_layernorm_fwd_static (the entry point that allocates nothing: its partial sums live in a device
array of the library image) against mgx_layernorm_fwd, through the C-ABI, on two calls in a row, the
second with a smaller input.  Same kernels, so the results must be bit-identical."""
import ctypes as C

import pytest
import torch

from music_generator_amd import _cabi

pytestmark = pytest.mark.gpu


def _run(lib, nimg, e, seed):
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    x = torch.randn((nimg, e), device="cuda", generator=g) * 3.0 + 0.5
    gamma = torch.rand((e,), device="cuda", generator=g) + 0.5
    beta = torch.randn((e,), device="cuda", generator=g)
    st = torch.cuda.current_stream().cuda_stream
    outs = []
    for static in (False, True):
        y = torch.empty_like(x)
        stats = torch.empty((nimg, 2), device="cuda")
        if static:
            rc = lib.mgx_layernorm_fwd_static(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(),
                                              stats.data_ptr(), nimg, e, 1e-3, st)
        else:
            nbytes = lib.mgx_layernorm_workspace(nimg, e)
            ws = torch.empty((nbytes,), dtype=torch.uint8, device="cuda")
            rc = lib.mgx_layernorm_fwd(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), stats.data_ptr(),
                                       nimg, e, 1e-3, ws.data_ptr(), nbytes, st)
        assert rc == _cabi.MGX_OK, rc
        torch.cuda.synchronize()
        outs.append((y, stats))
    (y0, s0), (y1, s1) = outs
    assert torch.equal(y0, y1) and torch.equal(s0, s1)
    # and the values are a LayerNormalization: per-sample mean / variance of (y - beta) / gamma
    z = ((y1 - beta) / gamma).double()
    assert float(z.mean(dim=1).abs().max()) < 1e-4
    v = x.double().var(dim=1, unbiased=False)
    assert torch.allclose(z.var(dim=1, unbiased=False), v / (v + 1e-3), rtol=1e-4)


def test_static_scratch_layernorm_matches_workspace_entry_point():
    lib = _cabi.load()
    _run(lib, 7, 8 * 1500 * 2, 1)      # 7 samples of an 8 x 1500 x 2 clip
    _run(lib, 3, 4 * 750 * 2, 2)       # smaller input afterwards: stale partial sums must not matter
    assert lib.mgx_layernorm_fwd_static_max_samples() == 2048
    x = torch.zeros((4,), device="cuda")
    rc = lib.mgx_layernorm_fwd_static(x.data_ptr(), x.data_ptr(), x.data_ptr(), x.data_ptr(), x.data_ptr(), 2049, 2,
                                      1e-3, None)
    assert rc == _cabi.MGX_ERR_WORKSPACE
