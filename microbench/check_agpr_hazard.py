#!/usr/bin/env python
"""Scan a gfx950 assembly listing for the two hazards of the inline-asm MFMAs (SNN_AGPR_MFMA, snn_kernels.h); the
compiler does not know the asm is a matrix instruction and inserts nothing for it:
  before: an AGPR written by v_accvgpr_write_b32 fewer than NEED wait states before a v_mfma reads it as its A operand;
  after:  a non-matrix instruction touching the accumulator of such a v_mfma fewer than AFTER wait states behind it
          (16-pass fp32 MFMA: 18; the kernels put `s_nop 15; s_nop 3` there).
Every instruction in between counts as one wait state, `s_nop N` as N + 1.
   make -C supernnova_amd/csrc check-hazard     or     python microbench/check_agpr_hazard.py file.s [NEED=2] [AFTER=18]"""
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
NEED = int(sys.argv[2]) if len(sys.argv) > 2 else 2
AFTER = int(sys.argv[3]) if len(sys.argv) > 3 else 18
bad = 0
sites = 0
closest = closest_after = None
kernel = None
recent = {}    # agpr index -> clock after its last v_accvgpr_write
pending = {}   # (lo, hi) accumulator range -> clock after the last AGPR-operand MFMA that wrote it
clock = 0
for l in lines:
    if re.match(r"^_Z\w+:", l):
        kernel = l.split(":")[0]; recent = {}; pending = {}; clock = 0
    t = l.split(";")[0].strip()
    if not t or t.startswith(".") or t.endswith(":"):
        continue
    mm = re.match(r"v_mfma_\w+ a\[(\d+):(\d+)\], (a?)(\d*)", t)
    if mm:
        rng = (int(mm.group(1)), int(mm.group(2)))
        if mm.group(3) == "a":
            sites += 1
            a = int(mm.group(4))
            if a in recent:
                gap = clock - recent[a]
                closest = gap if closest is None else min(closest, gap)
                if gap < NEED:
                    bad += 1
                    print(f"{kernel}: a{a} written {gap} wait states before the MFMA that reads it")
            pending[rng] = clock + 1
        else:
            pending.pop(rng, None)     # a matrix instruction the compiler knows: it places that one's wait states itself
    elif pending:
        used = set(int(x) for x in re.findall(r"\ba(\d+)\b", t))
        for lo, hi in re.findall(r"\ba\[(\d+):(\d+)\]", t):
            used.update(range(int(lo), int(hi) + 1))
        for rng in [r for r in pending if any(r[0] <= u <= r[1] for u in used)]:
            gap = clock - pending.pop(rng)
            closest_after = gap if closest_after is None else min(closest_after, gap)
            if gap < AFTER:
                bad += 1
                print(f"{kernel}: a[{rng[0]}:{rng[1]}] touched {gap} wait states after an AGPR-operand MFMA: {t}")
    m = re.match(r"s_nop (\d+)", t)
    clock += int(m.group(1)) + 1 if m else 1
    m = re.match(r"v_accvgpr_write_b32 a(\d+),", t)
    if m:
        recent[int(m.group(1))] = clock
print(f"{sites} AGPR-operand MFMAs; closest write {closest} wait states before its use (need {NEED}), "
      f"closest accumulator access {closest_after} behind one (need {AFTER}); {bad} hazards")
sys.exit(1 if bad else 0)
