#!/usr/bin/env python
"""Event sequence (loads, vmcnt waits, barriers, MFMA groups, LDS, stores, branches) of a line range of one kernel in a
gfx950 assembly listing: python microbench/isa_loop_events.py file.s <kernel substring> <first line> <last line>
(line numbers relative to the kernel label, as microbench/isa_loop_mix.py prints them)."""
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*" + re.escape(sys.argv[2]) + r"\w*:", l))
body = lines[start + int(sys.argv[3]):start + int(sys.argv[4]) + 1]
out, last, cnt = [], None, 0
for l in body + ["END"]:
    t = l.strip().split()
    if not t:
        continue
    k = None
    if t[0] == "END": k = "END"
    elif t[0].startswith("global_load"): k = "LD"
    elif t[0].startswith("global_store"): k = "ST"
    elif "mfma" in t[0]: k = "MFMA"
    elif t[0] == "s_barrier": k = "BAR"
    elif t[0] == "s_waitcnt" and "vmcnt" in l: k = "W" + re.search(r"vmcnt\((\d+)\)", l).group(1)
    elif t[0].startswith("ds_"): k = "lds"
    elif t[0].startswith("s_cbranch"): k = "BR"
    elif t[0].startswith("v_"): k = "v"
    else: continue
    if k != last:
        if last: out.append(f"{last}x{cnt}")
        last, cnt = k, 0
    cnt += 1
print(" ".join(out))
