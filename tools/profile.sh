#!/bin/bash
# Run ON THE GPU BOX (via gpurun): kernel trace + PMC passes for the default bench workload.
#   tools/profile.sh <tag>      -> gpurun_out/prof_<tag>/{kt,fetch,write,sq1,sq2}/...  + gpurun_out/prof_<tag>/summary.json
# PMC passes are separate runs, never combined with other trace domains (pool rule).
set -e
TAG=${1:-r01}
OUT=gpurun_out/prof_$TAG
BENCH="python3 bench.py --no-cpu-baseline --no-extras --steps 5 --warmup 2 ${BENCH_ARGS:-}"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
case "${BENCH_ARGS:-}" in *c5*) export PROF_BYTES_PER_READ=356;; esac   # -l 160: 2 * 160 + 4 + 32 algorithmic bytes per read
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -- $BENCH > "$OUT/kt.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- $BENCH > "$OUT/fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- $BENCH > "$OUT/write.log" 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS \
    --output-format csv -d "$OUT/sq1" -- $BENCH > "$OUT/sq1.log" 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR \
    --output-format csv -d "$OUT/sq2" -- $BENCH > "$OUT/sq2.log" 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$OUT/misc" -- $BENCH > "$OUT/misc.log" 2>&1 || true
python3 tools/summarize_prof.py "$OUT" demux_kernel > "$OUT/summary.json"
python3 tools/summarize_prof.py "$OUT" prescan_transpose_kernel > "$OUT/summary_prescan_transpose.json"
python3 tools/summarize_prof.py "$OUT" prescan_dp_kernel > "$OUT/summary_prescan_dp.json"
python3 tools/make_pmc_summary.py "$OUT" > "$OUT/pmc_summary.json"
cat "$OUT/pmc_summary.json"
