#!/usr/bin/env python3
"""Timeline of the launch stream around the exchanges of `bench.py --force-dist` from a rocprofv3 kernel trace (x_kernel_trace.csv):
for every pack / mailbox / RCCL kernel the kernels before and after it with start, end, duration and the idle gap in front (us).

    python tools/exchange_timeline.py gpurun_out/r05x/trace_c2/x_kernel_trace.csv [n_context]
"""
import csv
import sys


def main(path, ctx=12):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    hot = [i for i, r in enumerate(rows) if any(k in r["Kernel_Name"] for k in ("k_pack_sums", "k_mailbox", "nccl", "Nccl", "rccl"))]
    print(f"# {path}: {len(rows)} kernel records, {len(hot)} exchange-related")
    shown, last_end = set(), None
    for i in hot:
        lo, hi = max(0, i - ctx), min(len(rows), i + ctx + 1)
        if i in shown:
            continue
        t0 = int(rows[lo]["Start_Timestamp"])
        print(f"--- around record {i}")
        prev_end = None
        for j in range(lo, hi):
            r = rows[j]
            shown.add(j)
            s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
            gap = "" if prev_end is None else f"{(s - prev_end) / 1e3:7.1f}"
            prev_end = e
            print(f"{s / 1e3:9.1f} {e / 1e3:9.1f}  dur {(e - s) / 1e3:6.1f}  gap {gap:>7}  stream {r.get('Stream_Id', '?'):>2}  {r['Kernel_Name'][:70]}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 12)
