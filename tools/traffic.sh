#!/bin/bash
# quick HBM traffic check of the demux kernel (FETCH_SIZE / WRITE_SIZE, separate PMC passes) on the GPU box
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_$c
  rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc_$c -- python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 > /dev/null 2>&1
  python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(float)
for f in glob.glob("gpurun_out/pmc_$c/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        if "demux" in r["Kernel_Name"]: acc[r["Dispatch_Id"]]+=float(r["Counter_Value"])
v=sum(acc.values())/max(len(acc),1)
print("$c: %.1f MB per launch%s" % (v*1024*(2 if "$c"=="FETCH_SIZE" else 1)/1e6, " (x2 gfx950 correction applied)" if "$c"=="FETCH_SIZE" else ""))
PY
done
