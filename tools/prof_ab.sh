#!/bin/bash
# Run ON THE GPU BOX: demux kernel duration under rocprofv3 --kernel-trace for several builds (SMX_LIB).  tools/prof_ab.sh label=lib ...
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for spec in "$@"; do
  label=${spec%%=*}; lib=${spec#*=}
  if [ -n "$lib" ]; then export SMX_LIB=$PWD/$lib; else unset SMX_LIB; fi
  rm -rf gpurun_out/prof_ab_$label
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_ab_$label -- python3 bench.py --no-cpu-baseline --no-extras --steps 5 --warmup 2 --streams 1 ${BENCH_ARGS:-} > gpurun_out/prof_ab_$label.log 2>&1
  python3 - "$label" <<'PY'
import csv, glob, sys, json
label = sys.argv[1]
f = glob.glob(f"gpurun_out/prof_ab_{label}/*/*_kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if "smx::" in r["Name"]:
        print(label, r["Name"][:48], r["Calls"], "avg %.1f us" % (float(r["AverageNs"]) / 1e3))
for ln in open(f"gpurun_out/prof_ab_{label}.log"):
    if ln.startswith('{"metric"'):
        j = json.loads(ln); print(label, "bench in the same process: step %.4f ms" % j["ms_per_step"], [round(k["ms"], 4) for k in j["step_kernels"]])
PY
done
