#!/usr/bin/env python3
"""Folds the PMC passes of tools/prof_valu_busy.sh (gpurun_out/valu/) into profiles/r03_valu_busy.json: per kernel, the mean of
every SQ counter per dispatch, plus per-wavefront figures (SQ cycle counters count quad-cycles: x 4)."""
import csv
import json
import re
import sys
from collections import defaultdict
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
src = ROOT / "gpurun_out" / "valu"
out = {}
for d in sorted(src.iterdir()):
    if not d.is_dir():
        continue
    target = re.sub(r"_p\d+$", "", d.name)
    for f in d.rglob("*counter_collection.csv"):
        acc = defaultdict(lambda: defaultdict(list))
        with open(f) as fh:
            for r in csv.DictReader(fh):
                acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in acc.items():
            if k.startswith("void at::") or "rocclr" in k or "elementwise" in k:
                continue
            slot = out.setdefault(target, {}).setdefault(k, {})
            for c, v in cs.items():
                slot[c] = sum(v) / len(v)
            slot["dispatches"] = max(slot.get("dispatches", 0), max(len(v) for v in cs.values()))
for target, ks in out.items():
    for k, c in ks.items():
        w = c.get("SQ_WAVES")
        if not w:
            continue
        per = {}
        for name in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_FMA_F32",
                     "SQ_INSTS_VALU_TRANS_F32", "SQ_INSTS_VALU_CVT", "SQ_INSTS_VALU_INT32", "SQ_INSTS_VMEM_WR", "SQ_INSTS_VMEM_RD", "SQ_INSTS_SMEM"):
            if name in c:
                per[name.replace("SQ_INSTS_", "insts_").lower()] = round(c[name] / w, 1)
        for name in ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_VALU2", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY",
                     "SQ_ACTIVE_INST_SCA", "SQ_INST_CYCLES_SALU", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_MISC"):
            if name in c:
                per[name.replace("SQ_", "").lower() + "_cycles"] = round(4 * c[name] / w, 1)
        if "SQ_THREAD_CYCLES_VALU" in c:
            per["thread_cycles_valu"] = round(c["SQ_THREAD_CYCLES_VALU"] / w, 1)
        for name in ("SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_CYCLES", "SQ_BUSY_CU_CYCLES"):
            if name in c:
                per[name.lower()] = round(c[name], 1)
        c["per_wave"] = per
(ROOT / "profiles" / "r03_valu_busy.json").write_text(json.dumps(out, indent=1))
for target, ks in out.items():
    print("==", target)
    for k, c in ks.items():
        if "per_wave" in c:
            print("  ", k[:70], int(c["SQ_WAVES"]), json.dumps(c["per_wave"]))
