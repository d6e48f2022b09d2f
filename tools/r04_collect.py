#!/usr/bin/env python3
"""gpurun_out/r04 -> profiles/r04_*: copies the bench lines and kernel-stat tables, and folds the PMC passes into profiles/r04_pmc.json
(per workload: HBM traffic per launch with the guide's corrections calibrated in the same run -- the calibration kernel's RAW counter
values are recorded --, VALU / SALU / LDS instructions per wave of the dominant kernel)."""
import csv, json, shutil
from collections import defaultdict
from pathlib import Path
R = Path(__file__).resolve().parent.parent
O, P = R / "gpurun_out" / "r04", R / "profiles"


def mean_counter(d, kernel_sub):
    acc = defaultdict(list)
    for f in Path(d).rglob("*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if kernel_sub in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}


for f in O.glob("bench_*"):
    shutil.copy(f, P / ("r04_" + f.name))
for d in O.glob("prof_*"):
    for f in d.rglob("*kernel_stats.csv"):
        shutil.copy(f, P / f"r04_kernel_stats_{d.name[5:]}.csv")
calib_f, nf = mean_counter(O / "pmc" / "calib_FETCH_SIZE", "vectorized_elementwise_kernel")     # the 20 copies, not the initialising rand kernel
calib_w, nw = mean_counter(O / "pmc" / "calib_WRITE_SIZE", "vectorized_elementwise_kernel")
copy_bytes = 50331648
assert calib_f and calib_w, "the calibration kernel was not found in the counter files"
fc = copy_bytes / 1024 / calib_f["FETCH_SIZE"]          # how many bytes one counted KiB of reads stands for
wc = copy_bytes / 1024 / calib_w["WRITE_SIZE"]
out = {"note": "rocprofv3 PMC, separate passes (--kernel-trace --pmc X; tools/run_r04_profiles.sh), MI355X, 60 launches each of "
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
    "c4": ("c4:spheres:4096x64:specialized", "k_rollout", 540, 262144),
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
(P / "r04_pmc.json").write_text(json.dumps(out, indent=1))
print(json.dumps(out, indent=1))
