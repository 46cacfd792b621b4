#!/usr/bin/env python3
"""Summarise the rocprofv3 CSVs written by tools/profile.sh: per-launch averages for the demux kernel.
HBM traffic follows MI355X_MICROARCH.md section HBM: FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE
reports exactly half of the bytes of a wide coalesced (16 B/lane) read stream, so it is doubled; WRITE_SIZE is
taken as is."""
import csv
import glob
import json
import os
import sys

root = sys.argv[1]
KERNEL = sys.argv[2] if len(sys.argv) > 2 else "demux_kernel"   # kernel name substring (prescan_kernel for the primer prescan)


STEP_KERNEL = "prescan_transpose_kernel"   # launched exactly once per step: its dispatch count = the number of steps
READS = int(os.environ.get("PROF_READS", "765000"))
BYTES_PER_READ = int(os.environ.get("PROF_BYTES_PER_READ", "196"))   # 2 * search_len + 4 + 32: 196 at -l 80, 356 at -l 160


def pmc(sub):
    """{counter: value per STEP} from <root>/<sub>/**/*counter_collection.csv.  A step may launch the kernel more than once
    (compact demux launch + its redo launch: both match "demux_kernel"): their counters are summed.  Without the prescan
    (no step kernel in the trace) the mean per dispatch is returned."""
    acc, steps = {}, set()
    for path in glob.glob(os.path.join(root, sub, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as fh:
            for row in csv.DictReader(fh):
                if STEP_KERNEL in row.get("Kernel_Name", ""):
                    steps.add(row["Dispatch_Id"])
                if KERNEL not in row.get("Kernel_Name", ""):
                    continue
                acc.setdefault(row["Counter_Name"], {}).setdefault(row["Dispatch_Id"], 0.0)
                acc[row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
    return {k: sum(v.values()) / (len(steps) if steps else len(v)) for k, v in acc.items()}


out = {"root": root}
for path in glob.glob(os.path.join(root, "kt", "**", "*kernel_stats.csv"), recursive=True):
    with open(path) as fh:
        rows = list(csv.DictReader(fh))
    steps = sum(int(r["Calls"]) for r in rows if STEP_KERNEL in r["Name"])
    mine = [r for r in rows if KERNEL in r["Name"]]
    if mine:   # per step: the durations of all matching launches of a step added up
        calls = sum(int(r["Calls"]) for r in mine)
        total = sum(float(r["TotalDurationNs"]) for r in mine)
        out["kernel_trace"] = {"calls": calls, "launches_per_step": calls / steps if steps else 1.0,
                               "avg_ns": total / (steps if steps else calls),
                               "min_ns": min(float(r["MinNs"]) for r in mine), "max_ns": max(float(r["MaxNs"]) for r in mine)}
for path in glob.glob(os.path.join(root, "kt", "**", "*kernel_trace.csv"), recursive=True):
    with open(path) as fh:
        for row in csv.DictReader(fh):
            if KERNEL in row["Kernel_Name"]:
                out["dispatch"] = {k: row[k] for k in ("LDS_Block_Size", "Scratch_Size", "VGPR_Count", "Accum_VGPR_Count",
                                                       "SGPR_Count", "Workgroup_Size_X", "Grid_Size_X")}
                break
counters = {}
for sub in ("fetch", "write", "sq1", "sq2", "misc"):
    counters.update(pmc(sub))
out["counters_per_launch"] = counters
if "FETCH_SIZE" in counters and "WRITE_SIZE" in counters:
    fetch_b = counters["FETCH_SIZE"] * 1024 * 2      # gfx950: counter reads half of a 16 B/lane stream
    write_b = counters["WRITE_SIZE"] * 1024
    out["hbm"] = {"fetch_bytes_corrected": fetch_b, "write_bytes": write_b, "hbm_bytes_per_launch": fetch_b + write_b,
                  "reads_per_launch": READS, "algorithmic_bytes_per_launch": READS * BYTES_PER_READ}
print(json.dumps(out, indent=1))
