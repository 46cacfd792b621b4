#!/bin/bash
# Run ON THE GPU BOX: 8-primer panel, small compact tiles at forced workgroups per CU (what a further resident workgroup is worth).
#   tools/c3_residency_exp.sh "Rc blocks [lib]" ...      e.g.  "32 3" "32 4" "32 5 build_exp/libsmx_w5.so"
cfgs=("$@")
for rep in 1 2; do
for cfg in "${cfgs[@]}"; do set -- $cfg
  if [ -n "$3" ]; then export SMX_LIB=$PWD/$3; else unset SMX_LIB; fi
  out=$(SMX_COMPACT_R=$1 SMX_BLOCKS_PER_CU=$2 python3 bench.py --no-cpu-baseline --no-extras --config ${CFG:-c3} --reads 1000000 --steps 20 --warmup 3 --rotate 2 2>/dev/null)
  python3 -c "
import json,sys
d=json.loads(sys.argv[1]); print('Rc=$1 blocks=$2 lib=${3:-default}', 'step %.4f ms' % d['ms_per_step'], 'kernels', [round(k['ms'],4) for k in d['step_kernels']])" "$out"
done; done
