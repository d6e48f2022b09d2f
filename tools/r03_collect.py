#!/usr/bin/env python3
"""gpurun_out/r03 -> profiles/r03_*: copies the bench lines and kernel-stat tables, and folds the PMC passes into profiles/r03_pmc.json
(per workload: HBM traffic per launch with the guide's corrections calibrated in the same run, VALU / SALU / LDS instructions per wave)."""
import csv, json, shutil, sys
from collections import defaultdict
from pathlib import Path
R = Path(__file__).resolve().parent.parent
O, P = R / "gpurun_out" / "r03", R / "profiles"


def mean_counter(d, kernel_sub):
    acc = defaultdict(list)
    for f in Path(d).rglob("*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if kernel_sub in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}


for f in O.glob("bench_*"):
    shutil.copy(f, P / ("r03_" + f.name))
for d in O.glob("prof_*"):
    for f in d.rglob("*kernel_stats.csv"):
        shutil.copy(f, P / f"r03_kernel_stats_{d.name[5:]}.csv")
calib_f, _ = mean_counter(O / "pmc" / "calib_FETCH_SIZE", "elementwise")
calib_w, _ = mean_counter(O / "pmc" / "calib_WRITE_SIZE", "elementwise")
copy_bytes = 50331648
fc = copy_bytes / 1024 / calib_f["FETCH_SIZE"] if calib_f else 2.0          # how many bytes one counted KiB of reads stands for
wc = copy_bytes / 1024 / calib_w["WRITE_SIZE"] if calib_w else 1.0
out = {"note": "rocprofv3 PMC, separate passes (--kernel-trace --pmc X; tools/run_r03_profiles.sh), MI355X, 60 launches each of "
               "bench.py's default command per scene.  FETCH_SIZE / WRITE_SIZE are KiB; corrections per /opt/skills/guides/"
               "MI355X_MICROARCH.md (HBM section), calibrated in the same run on a 50 331 648-byte device copy "
               f"(FETCH x{fc:.3f}, WRITE x{wc:.3f}).  valu_insts_per_wave = SQ_INSTS_VALU / SQ_WAVES of the rollout kernel.",
       "calibration": {"copy_bytes": copy_bytes, "fetch_kib": calib_f.get("FETCH_SIZE"), "write_kib": calib_w.get("WRITE_SIZE")},
       "workloads": {}}
algo = {"spheres": 192, "grid": 272, "shelf": 192, "maze": 192, "c3": 192}
for s in ("spheres", "grid", "shelf", "maze", "c3"):
    f, _ = mean_counter(O / "pmc" / f"{s}_FETCH_SIZE", "k_rollout")
    w, _ = mean_counter(O / "pmc" / f"{s}_WRITE_SIZE", "k_rollout")
    q, _ = mean_counter(O / "pmc" / f"{s}_SQ_INSTS_VALU", "k_rollout")
    if not (f and w):
        continue
    key = ("c3:spheres" if s == "c3" else f"c2:{s}") + ":4096x64:specialized"
    rec = {"fetch_size_kib_raw": f["FETCH_SIZE"], "write_size_kib_raw": w["WRITE_SIZE"], "fetch_correction": fc, "write_correction": wc,
           "traffic_bytes_per_launch": int(1024 * (f["FETCH_SIZE"] * fc + w["WRITE_SIZE"] * wc)),
           "algorithmic_bytes_per_launch": algo[s] * 262144}
    if q and q.get("SQ_WAVES"):
        rec.update(valu_insts_per_wave=q["SQ_INSTS_VALU"] / q["SQ_WAVES"], salu_insts_per_wave=q["SQ_INSTS_SALU"] / q["SQ_WAVES"],
                   lds_insts_per_wave=q["SQ_INSTS_LDS"] / q["SQ_WAVES"])
    out["workloads"][key] = rec
(P / "r03_pmc.json").write_text(json.dumps(out, indent=1))
print(json.dumps(out["workloads"], indent=1))
