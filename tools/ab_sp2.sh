#!/bin/bash
# Run ON THE GPU BOX: the two-primer kernel at 64-read tiles x 4 workgroups per CU against an experiment build (SMX_EXP) with
# other tile sizes.   tools/ab_sp2.sh "label=lib=TILE_R" ...
run() {
  out=$(python3 bench.py --no-cpu-baseline --no-extras --config c2 --reads 765000 --steps 20 --warmup 3 --rotate 2 --streams ${STREAMS:-1} 2>/tmp/ab_err.txt)
  grep -m2 "lean R=\|phase timing" /tmp/ab_err.txt
  python3 -c "
import json,sys
d=json.loads(sys.argv[1]); print('$1', 'step %.4f ms' % d['ms_per_step'], 'kernels', [round(k['ms'],4) for k in d['step_kernels']])" "$out"
}
for rep in 1 2; do
for spec in "$@"; do
  IFS='=' read -r label lib tile <<< "$spec"
  if [ -n "$lib" ]; then export SMX_LIB=$PWD/$lib; else unset SMX_LIB; fi
  if [ -n "$tile" ]; then export SMX_TILE_R=$tile; else unset SMX_TILE_R; fi
  SMX_DEBUG=1 run "$label"
done
done
