#!/bin/bash
# VALU instructions of the demux kernel with whole phases compiled out (build_exp/libsmx_<n>.so from `make exp EXP=<n>`:
# 6 = no barcode work, 7 = no primer columns): the differences are the phases' instruction budgets.  Run ON THE GPU BOX.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in base 6 7; do
  lib=""; [ "$v" != base ] && lib="$GRAFT_REPO_ROOT/build_exp/libsmx_$v.so"
  rm -rf gpurun_out/valu_$v
  SMX_LIB=$lib rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU --output-format csv -d gpurun_out/valu_$v -- python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/valu_$v.log 2>&1
  python3 - "$v" <<'PY'
import csv,glob,collections,sys
acc=collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("gpurun_out/valu_%s/**/*counter_collection.csv" % sys.argv[1],recursive=True):
    for r in csv.DictReader(open(f)):
        if "demux" in r["Kernel_Name"]: acc[r["Counter_Name"]][r["Dispatch_Id"]]+=float(r["Counter_Value"])
print(sys.argv[1], {k: "%.4g" % (sum(v.values())/max(len(v),1)) for k,v in acc.items()})
PY
done
