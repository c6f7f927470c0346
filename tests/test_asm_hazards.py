"""Static check of the kernels that issue MFMAs from inline assembly: no vector-ALU instruction writes a register that
an MFMA reads fewer than two wait states later (the compiler pads that distance for its own MFMAs only).  The scan
compiles the source to gfx950 ISA on the host; no GPU is needed."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not present")
@pytest.mark.parametrize("source,flags", [("gblock_bf16.hip", ["-mllvm", "-amdgpu-mfma-vgpr-form"]), ("bchain.hip", [])])
def test_no_valu_write_right_before_an_asm_mfma(source, flags):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "asm_mfma_hazard_scan.py"),
                          os.path.join(ROOT, "music_generator_amd", "csrc", source)] + flags,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    last = out.stdout.strip().splitlines()[-1]
    assert " inside asm statements), 0 with " in last and "(0 inside" not in last, out.stdout[-2000:]
