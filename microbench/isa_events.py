#!/usr/bin/env python
"""Event sequence of one kernel in a gfx950 assembly listing (hipcc -S --cuda-device-only), split at its barriers:
global loads / stores, s_waitcnt (v = vmcnt, l = lgkmcnt), MFMA runs, LDS reads / writes, VALU runs, branches.
   python microbench/isa_events.py file.s <substring of the mangled kernel name> [section numbers ...]"""
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*" + re.escape(sys.argv[2]) + r"\w*:", l))
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))


def kind(l):
    t = l.strip().split()
    if not t or t[0].startswith((";", ".")):
        return None
    if t[0].startswith("global_load") or t[0].startswith("buffer_load"): return "LD"
    if t[0].startswith("global_store") or t[0].startswith("buffer_store"): return "ST"
    if t[0].startswith("scratch_"): return "SCRATCH"
    if "mfma" in t[0]: return "MFMA"
    if t[0] == "s_barrier": return "BAR"
    if t[0] == "s_waitcnt":
        v, g = re.search(r"vmcnt\((\d+)\)", l), re.search(r"lgkmcnt\((\d+)\)", l)
        return "W" + ("v" + v.group(1) if v else "") + ("l" + g.group(1) if g else "")
    if t[0].startswith("ds_read") or t[0].startswith("ds_load"): return "dsR"
    if t[0].startswith("ds_write") or t[0].startswith("ds_store"): return "dsW"
    if t[0].startswith("s_cbranch") or t[0] == "s_branch": return "BR"
    if t[0].startswith("v_"): return "v"
    return None


out, last, cnt = [], None, 0
for l in lines[start:end] + ["END"]:
    k = "END" if l == "END" else kind(l)
    if k is None:
        continue
    if k != last:
        if last:
            out.append(f"{last}x{cnt}" if cnt > 1 else last)
        last, cnt = k, 0
    cnt += 1
parts = " ".join(out).split(" BAR ")
print(f"{end - start} lines, {len(parts) - 1} barriers; MFMA instructions per section:",
      [sum(int(m.group(1) or 1) for m in re.finditer(r"MFMA(?:x(\d+))?", p)) for p in parts])
for i in map(int, sys.argv[3:]):
    if i < len(parts):
        print(f"--- section {i}\n{parts[i]}")
