#!/usr/bin/env python3
"""Summarise a rocprofv3 counter_collection.csv: mean counter value per kernel launch.
usage: pmc_summary.py <counter_collection.csv> [kernel-name-substring]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
pat = sys.argv[2] if len(sys.argv) > 2 else ""
acc = defaultdict(list)
for r in rows:
    if pat in r["Kernel_Name"]:
        acc[(r["Kernel_Name"][:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(acc.items()):
    print(f"{k:60s} {c:28s} n={len(v):3d} mean={sum(v) / len(v):.6g}")
