#!/usr/bin/env python3
"""Merge the per-kernel summaries written by tools/profile.sh into the one file bench.py replays
(profiles/rNN_pmc_summary.json): wave64 VALU instructions and HBM bytes per STEP (= the three kernels of one
smx_batch_run_device call), per-kernel durations from rocprofv3 --kernel-trace --stats."""
import json
import os
import sys

root = sys.argv[1]
names = {"demux_kernel": "summary.json", "prescan_transpose_kernel": "summary_prescan_transpose.json",
         "prescan_dp_kernel": "summary_prescan_dp.json"}
READS = int(os.environ.get("PROF_READS", "765000"))
BYTES_PER_READ = int(os.environ.get("PROF_BYTES_PER_READ", "196"))   # 2 * search_len + 4 + 32: 196 at -l 80, 356 at -l 160
out = {"root": root, "reads_per_launch": READS, "kernels": {}}
valu = hbm = 0.0
ok = True
for k, f in names.items():
    path = os.path.join(root, f)
    if not os.path.exists(path):
        continue
    d = json.load(open(path))
    c = d.get("counters_per_launch", {})
    entry = {"avg_ns": d.get("kernel_trace", {}).get("avg_ns"), "dispatch": d.get("dispatch"),
             "counters_per_launch": c, "hbm": d.get("hbm")}
    out["kernels"][k] = entry
    if "SQ_INSTS_VALU" in c:
        valu += c["SQ_INSTS_VALU"]
    else:
        ok = False
    if d.get("hbm"):
        hbm += d["hbm"]["hbm_bytes_per_launch"]
    else:
        ok = False
out["valu_instr_per_step"] = valu if ok else None
out["valu_instr_by_kernel"] = {k: v["counters_per_launch"].get("SQ_INSTS_VALU") for k, v in out["kernels"].items()}
out["hbm_bytes_per_step"] = hbm if ok else None
out["algorithmic_bytes_per_step"] = READS * BYTES_PER_READ
print(json.dumps(out, indent=1))
