#!/usr/bin/env python3
"""Compact view of a rocprofv3 *_kernel_stats.csv: name (truncated), calls, average us, total ms, %."""
import csv
import sys
for path in sys.argv[1:]:
    print("==", path)
    for r in csv.DictReader(open(path)):
        print(f"{r['Name'][:78]:78s} {int(r['Calls']):6d} {float(r['AverageNs']) / 1e3:10.2f} us {float(r['TotalDurationNs']) / 1e6:9.3f} ms {float(r['Percentage']):6.2f}%")
