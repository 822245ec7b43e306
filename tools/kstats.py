#!/usr/bin/env python3
"""Prints calls and mean duration per kernel from a rocprofv3 *kernel_stats.csv."""
import csv
import glob
import re
import sys

for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(k_\w+|rocprim\w*::\w+|\w+Buffer\w*)", r["Name"])
        name = m.group(1) if m else r["Name"][:40]
        if len(sys.argv) < 3 or re.search(sys.argv[2], name):
            print(f"{name:28s} calls {int(r['Calls']):5d}  mean {float(r['AverageNs']) / 1e3:10.1f} us  total {float(r['TotalDurationNs']) / 1e6:8.2f} ms")
