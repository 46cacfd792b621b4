#!/bin/bash
# A/B of instruction counts per kernel: tools/ab_instr.sh <label>=<libsmx path or ""> ...   (GPU box)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for spec in "$@"; do
  L=${spec%%=*}; lib=${spec#*=}
  if [ -n "$lib" ]; then export SMX_LIB=$GRAFT_REPO_ROOT/$lib; else unset SMX_LIB; fi
  rm -rf gpurun_out/ab_$L
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d gpurun_out/ab_$L -- python3 bench.py --no-cpu-baseline --no-extras --steps 5 --warmup 2 ${BENCH_ARGS:-} > gpurun_out/ab_$L.log 2>&1
  python3 - "$L" <<'PY'
import csv, glob, collections, sys
L = sys.argv[1]
f = glob.glob(f"gpurun_out/ab_{L}/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"][:48]; acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_INSTS_VALU": n[k] += 1
for k in acc:
    if "smx" in k: print(L, k, n[k], {c: round(v / n[k] / 1e6, 2) for c, v in acc[k].items()})
PY
done
