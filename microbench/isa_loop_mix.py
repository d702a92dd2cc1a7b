#!/usr/bin/env python
"""Instruction mix of the innermost loop of one kernel in a gfx950 assembly listing (hipcc -S --cuda-device-only):
   python microbench/isa_loop_mix.py file.s <substring of the mangled kernel name>"""
import collections
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*" + re.escape(sys.argv[2]) + r"\w*:", l))
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
body = lines[start:end]
labels = {l.split(":")[0]: i for i, l in enumerate(body) if l.startswith(".LBB")}
best = None
for i, l in enumerate(body):
    m = re.search(r"s_cbranch_\w+\s+(\.LBB\w+)", l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:          # backward branch: a loop [label, i]
        a = labels[m.group(1)]
        if not any(re.search(r"s_cbranch_\w+\s+(\.LBB\w+)", body[j]) and labels.get(re.search(r"(\.LBB\w+)", body[j]).group(1), 1e9) < j
                   and labels.get(re.search(r"(\.LBB\w+)", body[j]).group(1), -1) > a for j in range(a, i)):
            if best is None or i - a > best[1] - best[0]:
                best = (a, i)
a, b = best
mix = collections.Counter()
for l in body[a:b + 1]:
    t = l.strip().split()
    if t and not t[0].startswith((";", ".")):
        mix[re.sub(r"_e(32|64)$", "", t[0])] += 1
print(f"loop lines {a}..{b} of kernel at {start}; {sum(mix.values())} instructions")
cat = collections.Counter()
for k, v in mix.items():
    c = ("mfma" if "mfma" in k else "accvgpr" if "accvgpr" in k else "vmem" if k.startswith(("global_", "buffer_", "flat_"))
         else "lds" if k.startswith("ds_") else "salu" if k.startswith("s_") else "valu")
    cat[c] += v
print(dict(cat))
print(", ".join(f"{k} {v}" for k, v in mix.most_common(24)))
