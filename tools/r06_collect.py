#!/usr/bin/env python3
"""gpurun_out/r06 -> profiles/r06_*: copies the bench lines and kernel-stat tables, and folds the PMC passes into profiles/r06_pmc.json
(per workload: HBM traffic per launch with the guide's corrections calibrated in the same run -- the calibration kernel's RAW counter
values are recorded --, VALU / SALU / LDS instructions per wave of the dominant kernel)."""
import csv, json, shutil
from collections import defaultdict
from pathlib import Path
R = Path(__file__).resolve().parent.parent
O, P = R / "gpurun_out" / "r06", R / "profiles"


def mean_counter(d, kernel_sub):
    acc = defaultdict(list)
    for f in Path(d).rglob("*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if kernel_sub in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}


for f in O.glob("bench_*"):
    shutil.copy(f, P / ("r06_" + f.name))
for d in O.glob("prof_*"):
    for f in d.rglob("*kernel_stats.csv"):
        shutil.copy(f, P / f"r06_kernel_stats_{d.name[5:]}.csv")
calib_f, nf = mean_counter(O / "pmc" / "calib_FETCH_SIZE", "vectorized_elementwise_kernel")     # the 20 copies, not the initialising rand kernel
calib_w, nw = mean_counter(O / "pmc" / "calib_WRITE_SIZE", "vectorized_elementwise_kernel")
copy_bytes = 50331648
assert calib_f and calib_w, "the calibration kernel was not found in the counter files"
fc = copy_bytes / 1024 / calib_f["FETCH_SIZE"]          # how many bytes one counted KiB of reads stands for
wc = copy_bytes / 1024 / calib_w["WRITE_SIZE"]
out = {"note": "rocprofv3 PMC, separate passes (--kernel-trace --pmc X; tools/run_r06_profiles.sh), MI355X, 60 launches each of "
               "bench.py per workload.  FETCH_SIZE / WRITE_SIZE are KiB; corrections per /opt/skills/guides/MI355X_MICROARCH.md (HBM "
               "section), calibrated in the same run on an elementwise kernel that reads and writes 50 331 648 bytes "
               f"(FETCH x{fc:.3f}, WRITE x{wc:.3f}; raw counter values under `calibration`).  valu_insts_per_wave = SQ_INSTS_VALU / "
               "SQ_WAVES of the workload's dominant kernel.",
       "calibration": {"kernel": "torch.mul(x, 1.0, out=y), 12 582 912 floats", "bytes_read": copy_bytes, "bytes_written": copy_bytes,
                       "fetch_size_kib_raw": calib_f["FETCH_SIZE"], "write_size_kib_raw": calib_w["WRITE_SIZE"],
                       "launches_averaged": [nf.get("FETCH_SIZE"), nw.get("WRITE_SIZE")], "fetch_correction": fc, "write_correction": wc},
       "workloads": {}}
W = {  # pass directory prefix -> (bench key, dominant kernel substring, algorithmic bytes per sample, samples per launch)
    "spheres": ("c2:spheres:4096x64:specialized", "k_rollout", 192, 262144),
    "grid": ("c2:grid:4096x64:specialized", "k_rollout", 272, 262144),
    "gridsmooth": ("c2:grid:4096x64:specialized:smooth", "k_rollout", 272, 262144),
    "shelf": ("c2:shelf:4096x64:specialized", "k_rollout", 192, 262144),
    "maze": ("c2:maze:4096x64:specialized", "k_rollout", 192, 262144),
    "c3": ("c3:spheres:4096x64:specialized", "k_rollout", 192, 262144),
    "c4": ("c4:spheres:4096x64:specialized", "k_rollout", 1096, 262144),       # one launch: rollout (540) + Jacobian outputs (556)
    "c5": ("c5:spheres:2048x128:specialized", "k_rollout_gpt", 254, 262144),
}
for s, (key, ksub, bps, n) in W.items():
    f, _ = mean_counter(O / "pmc" / f"{s}_FETCH_SIZE", ksub)
    w, _ = mean_counter(O / "pmc" / f"{s}_WRITE_SIZE", ksub)
    q, _ = mean_counter(O / "pmc" / f"{s}_SQ_INSTS_VALU", ksub)
    if not (f and w):
        continue
    rec = {"kernel": ksub, "fetch_size_kib_raw": f["FETCH_SIZE"], "write_size_kib_raw": w["WRITE_SIZE"],
           "traffic_bytes_per_launch": int(1024 * (f["FETCH_SIZE"] * fc + w["WRITE_SIZE"] * wc)),
           "algorithmic_bytes_per_launch": bps * n}
    rec["traffic_over_algorithmic"] = rec["traffic_bytes_per_launch"] / rec["algorithmic_bytes_per_launch"]
    if q and q.get("SQ_WAVES"):
        rec.update(valu_insts_per_wave=q["SQ_INSTS_VALU"] / q["SQ_WAVES"], salu_insts_per_wave=q["SQ_INSTS_SALU"] / q["SQ_WAVES"],
                   lds_insts_per_wave=q["SQ_INSTS_LDS"] / q["SQ_WAVES"])
    out["workloads"][key] = rec
# carried over from ROUND 5 (not re-measured): the same counters calibrated on the voxel-grid scene's ACCESS PATTERN -- independent random
# 16-byte (8-byte) gathers, one per 128-byte line of a 4 GiB table -- and on a coalesced stream of the same bytes (tools/gather_calib.hip;
# profiles/r05_gather_calibration.txt)
out["calibration_gather"] = {
    "tool": "tools/gather_calib.hip, 67 108 864 gathers per launch, table 4 GiB (measured in round 5, profiles/r05_gather_calibration.txt)",
    "gather_16B": {"algorithmic_bytes": 1073741824, "fetch_size_kib_raw": 4194336.8, "TCC_EA0_RDREQ_sum": 67109382.4, "TCC_EA0_RDREQ_32B_sum": 0.0,
                   "TCC_BUBBLE_sum": 0.0, "gathers_per_s": 47.03e9},
    "gather_8B": {"algorithmic_bytes": 536870912, "fetch_size_kib_raw": 4194317.0, "TCC_EA0_RDREQ_sum": 67109072.0, "TCC_EA0_RDREQ_32B_sum": 0.0,
                  "TCC_BUBBLE_sum": 0.0, "gathers_per_s": 47.09e9},
    "stream_16B": {"algorithmic_bytes": 1073741824, "fetch_size_kib_raw": 524300.5, "TCC_EA0_RDREQ_sum": 8388808.0, "gathers_per_s": 320.4e9},
    "reading": "one read request per gather, none of the 32-byte class; a request carries 128 bytes (the stream: 1 GiB in 8 388 808 requests) and "
               "FETCH_SIZE counts it as 64: the x 1.999 correction holds for gathers too -- a 16-byte (or 8-byte) random gather that misses the "
               "caches costs one 128-byte line (47.0 G requests/s x 128 B = 6.0 TB/s, what this HBM sustains)",
    "gathers_per_s_by_table_MiB": {"32": 204.6e9, "64": 56.3e9, "128": 54.5e9, "512": 48.0e9, "4096": 47.0e9}}
(P / "r06_pmc.json").write_text(json.dumps(out, indent=1))
print(json.dumps(out, indent=1))
