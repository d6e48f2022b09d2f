#!/usr/bin/env python3
"""Offline analysis of gpurun_out/phase_stamps.npy (tools/phase_profile.py): chip-wide timeline of the fused kernel.
Slot 2 is s_memrealtime at entry (100 MHz, chip-wide); all other slots are s_memtime (2.39 GHz, per-CU domain)."""
import sys
import numpy as np
r = np.load(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/phase_stamps.npy").astype(np.uint64)
hw = ((r[:, 2] >> np.uint64(44)) & np.uint64(0xffff)).astype(np.int64)
xcc = (r[:, 2] >> np.uint64(60)).astype(np.int64)
r[:, 2] &= np.uint64((1 << 44) - 1)
r = r.astype(np.int64)
simd, cu, sh, se = (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7
GHZ = 2.39
n = r.shape[0]
entry_us = (r[:, 2] - r[:, 2].min()) / 100.0                      # chip-wide entry time, us (10 ns resolution)
life_us = (r[:, 7] - r[:, 0]) / (GHZ * 1e3)
exit_us = entry_us + life_us
names = ["entry", "q loaded", "-", "pos staged", "objects done", "objectives done", "reverse done", "exit"]
print(f"waves {n}; entry: p5 {np.percentile(entry_us,5):.2f} p50 {np.percentile(entry_us,50):.2f} p95 {np.percentile(entry_us,95):.2f} max {entry_us.max():.2f} us")
print(f"wave lifetime: mean {life_us.mean():.2f} p5 {np.percentile(life_us,5):.2f} p95 {np.percentile(life_us,95):.2f} max {life_us.max():.2f} us")
print(f"exit: p5 {np.percentile(exit_us,5):.2f} p50 {np.percentile(exit_us,50):.2f} p95 {np.percentile(exit_us,95):.2f} max {exit_us.max():.2f} us   (kernel span first entry -> last exit)")
for k in (1, 3, 4, 5, 6, 7):
    t = entry_us + (r[:, k] - r[:, 0]) / (GHZ * 1e3)
    d = (r[:, k] - r[:, [0, 0, 0, 1, 3, 4, 5, 6][k]]) / (GHZ * 1e3)
    print(f"  {names[k]:16s} at p5 {np.percentile(t,5):5.2f} p50 {np.percentile(t,50):5.2f} p95 {np.percentile(t,95):5.2f} max {t.max():5.2f} us | phase duration mean {d.mean():.2f} us")
wg = np.arange(n) // 4
for x in range(8):
    if not ((wg % 8) == x).any():
        continue
    m = (wg % 8) == x
    print(f"  xcd{x}: entry p50 {np.percentile(entry_us[m],50):.2f} max {entry_us[m].max():.2f} | exit p50 {np.percentile(exit_us[m],50):.2f} max {exit_us[m].max():.2f} | life mean {life_us[m].mean():.2f}")
# occupancy over time: waves alive at t
ts = np.linspace(0, exit_us.max(), 25)
alive = [(int(((entry_us <= t) & (exit_us > t)).sum())) for t in ts]
print("alive waves over time:", " ".join(f"{t:.1f}:{a}" for t, a in zip(ts, alive)))

print("xcc ids of wg%8 groups:", [sorted(set(xcc[(wg % 8) == x].tolist())) for x in range(8)])
cuid = xcc * 1000 + se * 100 + sh * 50 + cu
simdid = cuid * 4 + simd
u, cnt = np.unique(cuid, return_counts=True)
print(f"distinct CUs used: {len(u)}; waves per CU histogram: {dict(zip(*np.unique(cnt, return_counts=True)))}")
u2, cnt2 = np.unique(simdid, return_counts=True)
print(f"distinct SIMDs used: {len(u2)}; waves per SIMD histogram: {dict(zip(*np.unique(cnt2, return_counts=True)))}")
# per-SIMD busy time = last exit - first entry; lifetime vs number of waves on the SIMD
busy = np.array([exit_us[simdid == s_].max() - entry_us[simdid == s_].min() for s_ in u2])
print(f"per-SIMD busy: mean {busy.mean():.2f} p5 {np.percentile(busy,5):.2f} p95 {np.percentile(busy,95):.2f} max {busy.max():.2f} us")
for c in sorted(set(cnt2.tolist())):
    sel = np.isin(simdid, u2[cnt2 == c])
    print(f"  SIMDs with {c} waves: {int((cnt2 == c).sum())} SIMDs, wave lifetime mean {life_us[sel].mean():.2f} max {life_us[sel].max():.2f}; busy mean {busy[cnt2 == c].mean():.2f}")
