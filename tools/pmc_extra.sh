#!/bin/bash
# extra PMC passes (instruction cache, branches, thread-level VALU utilisation) for the default bench workload;
# run ON THE GPU BOX.  Each pass is its own rocprofv3 run (pool rule: --pmc never combined with trace domains).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
BENCH="python3 bench.py --no-cpu-baseline --steps 3 --warmup 1"
i=0
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" \
           "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_BRANCH SQ_INSTS_SMEM" \
           "SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CU_CYCLES" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INST_CYCLES_SALU" \
           "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_EXP_GDS"; do
  i=$((i+1))
  rm -rf gpurun_out/pmcx_$i
  rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmcx_$i -- $BENCH > gpurun_out/pmcx_$i.log 2>&1
  python3 - "$i" <<'PY'
import csv,glob,collections,sys
acc=collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("gpurun_out/pmcx_%s/**/*counter_collection.csv" % sys.argv[1],recursive=True):
    for r in csv.DictReader(open(f)):
        if "demux" in r["Kernel_Name"]: acc[r["Counter_Name"]][r["Dispatch_Id"]]+=float(r["Counter_Value"])
for k,v in acc.items(): print("%s: %.4g per launch" % (k, sum(v.values())/max(len(v),1)))
PY
done
