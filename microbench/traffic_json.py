#!/usr/bin/env python
"""profiles/r3/traffic.json from the FETCH_SIZE / WRITE_SIZE passes of microbench/profile_round3.sh.

  python microbench/traffic_json.py <dir with pmc_<workload>_{FETCH_SIZE,WRITE_SIZE}/ and pmc_<workload>.json> > traffic.json

Per workload: the forward kernel (the dispatch with the largest FETCH_SIZE), its two counters summed over the dispatch's
rows (one row per XCD instance), corrected as MI355X_MICROARCH.md's HBM section prescribes: both counters are in KB;
FETCH_SIZE tallies 128-byte requests at 64 bytes on gfx950 and is doubled."""
import csv
import glob
import json
import sys
from collections import defaultdict

root = sys.argv[1]
out = {"command": "rocprofv3 --pmc FETCH_SIZE | --pmc WRITE_SIZE (separate passes) --output-format csv -- python3 bench.py "
                  "--workload W --steps 1 --warmup 0 --cpu-curves 0 --no-selfcheck --no-fast-path --no-end-to-end "
                  "--no-secondary --no-train (microbench/profile_round3.sh)",
       "note": "gfx950: FETCH_SIZE tallies 128-B requests at 64 B -> doubled (MI355X_MICROARCH.md, HBM). Counts L2<->fabric "
               "requests; Infinity-Cache hits are included, so this is an upper bound on HBM bytes.",
       "fetch_correction": 2.0, "workloads": {}}


def per_dispatch(d):
    acc = defaultdict(float)
    name = {}
    for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Dispatch_Id"]] += float(r["Counter_Value"])
            name[r["Dispatch_Id"]] = r["Kernel_Name"]
    return acc, name


for meta in sorted(glob.glob(f"{root}/pmc_*.json")):
    wl = meta[len(root) + 5:-5]
    line = json.load(open(meta))
    kname = line["roofline"]["kernel"]
    key = kname.split("<")[0].split(" ")[0]
    fetch, fname = per_dispatch(f"{root}/pmc_{wl}_FETCH_SIZE")
    write, wname = per_dispatch(f"{root}/pmc_{wl}_WRITE_SIZE")
    fsel = {k: v for k, v in fetch.items() if key in fname[k]}
    wsel = {k: v for k, v in write.items() if key in wname[k]}
    if not fsel or not wsel:
        print(f"{wl}: no dispatch of {key}", file=sys.stderr)
        continue
    # (one timed launch per pass: --steps 1 --warmup 0; should several dispatches match, take the mean)
    fkb, wkb = sum(fsel.values()) / len(fsel), sum(wsel.values()) / len(wsel)
    cfg = line["config"]
    out["workloads"][wl] = {
        "workload": {"curves_per_gpu": cfg["curves_per_gpu"], "samples": cfg["samples"], "max_len": cfg["max_len"]},
        "precision": "f32", "kernel": kname, "dispatches": len(fsel),
        "FETCH_SIZE_KB": fkb, "WRITE_SIZE_KB": wkb,
        "traffic_bytes_per_launch": (2.0 * fkb + wkb) * 1024.0,
        "algorithmic_flop_per_launch": line["roofline"]["flop_per_launch"],
    }
print(json.dumps(out, indent=1))
