#!/usr/bin/env python3
"""Scan the gfx950 ISA of a HIP source for a vector-ALU instruction that writes a register which an MFMA one or two
instructions later reads (the two wait states between a VALU write and an MFMA read of the same register are padded by
the compiler for MFMAs it emits itself, never inside an inline-assembly statement):
    python tools/asm_mfma_hazard_scan.py music_generator_amd/csrc/gblock_bf16.hip [extra hipcc flags]"""
import re, subprocess, sys, tempfile, os
src, extra = sys.argv[1], sys.argv[2:]
out = os.path.join(tempfile.mkdtemp(), "k.s")
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-S", "--cuda-device-only", src, "-o", out]
               + extra, check=True, stderr=subprocess.DEVNULL)
def regs(tok):
    m = re.match(r'([va])\[(\d+):(\d+)\]', tok)
    if m: return {(m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1)}
    m = re.match(r'([va])(\d+)$', tok)
    return {(m.group(1), int(m.group(2)))} if m else set()
real, in_asm = [], False
for l in open(out):
    l = l.strip()
    if l.startswith(";;#ASMSTART"): in_asm = True; continue
    if l.startswith(";;#ASMEND"): in_asm = False; continue
    if not l or l.startswith((';', '.')) or l.endswith(':'): continue
    real.append((l, in_asm))
total = asm_total = bad = 0
for i, (l, a) in enumerate(real):
    if not l.startswith('v_mfma'): continue
    total += 1; asm_total += a
    ops = [t.strip() for t in l.split(None, 1)[1].split(',')]
    src_regs = set().union(*[regs(t) for t in ops[1:]])
    states = 0
    for back in range(1, 4):
        p = real[i - back][0]
        if p.startswith('s_nop'): states += int(p.split()[1]) + 1; continue
        if states >= 2: break
        if p.startswith('v_') and not p.startswith('v_mfma') and regs(p.split(None, 1)[1].split(',')[0].strip()) & src_regs:
            bad += 1; print("  %s  ->  %s%s" % (p, l, "   [asm]" if a else ""))
        states += 1
print("%s: %d MFMAs (%d inside asm statements), %d with a VALU write of an operand fewer than 2 states ahead" % (src, total, asm_total, bad))
