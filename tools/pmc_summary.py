#!/usr/bin/env python3
"""Summarise rocprofv3 counter_collection CSVs: mean counter value per kernel launch, for kernels matching a substring.
usage: python tools/pmc_summary.py <dir with */p_counter_collection.csv> [kernel substring]"""
import csv
import sys
from collections import defaultdict
from pathlib import Path
root = Path(sys.argv[1])
pat = sys.argv[2] if len(sys.argv) > 2 else "k_rollout"
for f in sorted(root.rglob("*counter_collection.csv")):
    acc = defaultdict(list)
    with open(f) as fh:
        for r in csv.DictReader(fh):
            if pat in r["Kernel_Name"] or (pat == "k_rollout" and "elementwise" in r["Kernel_Name"] and "copy" in r["Kernel_Name"].lower()):
                acc[(r["Kernel_Name"][:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in sorted(acc.items()):
        print(f"{f.parent.name:28s} {k:60s} {c:28s} n={len(v):4d} mean={sum(v)/len(v):14.2f}")
