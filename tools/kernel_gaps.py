#!/usr/bin/env python3
"""From a rocprofv3 --kernel-trace CSV: mean duration of a kernel and mean gap between consecutive launches of it."""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if sys.argv[2] in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[len(rows) // 4:]                       # steady state: drop the first quarter
dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows]
gap = [int(b["Start_Timestamp"]) - int(a["End_Timestamp"]) for a, b in zip(rows, rows[1:])]
gap = [g for g in gap if g < 100000]               # back-to-back launches only
print(f"{sys.argv[2]}: n={len(rows)}  duration mean {sum(dur)/len(dur)/1e3:.2f} us  gap mean {sum(gap)/max(1,len(gap))/1e3:.2f} us  period {(sum(dur)/len(dur)+sum(gap)/max(1,len(gap)))/1e3:.2f} us")
