#!/usr/bin/env python
"""Scan a gfx950 assembly listing for the hazard of the inline-asm MFMAs (SNN_AGPR_MFMA, snn_kernels.h): an AGPR written
by v_accvgpr_write_b32 fewer than NEED wait states before a v_mfma that reads it as its A operand.  The compiler does not
know the asm is a matrix instruction and inserts nothing there itself.  Every instruction between the two counts as one
wait state, `s_nop N` as N + 1.
   hipcc ... -S -o file.s file.hip && python microbench/check_agpr_hazard.py file.s [NEED=2]"""
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
NEED = int(sys.argv[2]) if len(sys.argv) > 2 else 2
bad = 0
sites = 0
closest = None
kernel = None
recent = {}   # agpr index -> wait-state clock at its last v_accvgpr_write
clock = 0
for l in lines:
    if re.match(r"^_Z\w+:", l):
        kernel = l.split(":")[0]; recent = {}; clock = 0
    t = l.strip()
    if not t or t.startswith((";", ".")) or t.endswith(":"):
        continue
    m = re.match(r"v_mfma_\w+ a\[\d+:\d+\], a(\d+),", t)
    if m:
        sites += 1
        a = int(m.group(1))
        if a in recent:
            gap = clock - recent[a]
            closest = gap if closest is None else min(closest, gap)
            if gap < NEED:
                bad += 1
                print(f"{kernel}: a{a} written {gap} wait states before the MFMA that reads it")
    m = re.match(r"s_nop (\d+)", t)
    clock += int(m.group(1)) + 1 if m else 1
    m = re.match(r"v_accvgpr_write_b32 a(\d+),", t)
    if m:
        recent[int(m.group(1))] = clock
print(f"{sites} AGPR-operand MFMAs, closest write {closest} wait states before its use, {bad} hazards (need {NEED})")
sys.exit(1 if bad else 0)
