#!/bin/bash
# Run ON THE GPU BOX: the kernel-resident bench under several builds of the library (SMX_LIB), same process order twice.
#   tools/ab_libs.sh cfgs "label=path-or-empty" ...
cfgs=$1; shift
for rep in 1 2; do
for spec in "$@"; do
  label=${spec%%=*}; lib=${spec#*=}
  for cfg in $cfgs; do
    reads=765000; [ $cfg != c2 ] && reads=1000000
    if [ -n "$lib" ]; then export SMX_LIB=$PWD/$lib; else unset SMX_LIB; fi
    out=$(python3 bench.py --no-cpu-baseline --no-extras --config $cfg --reads $reads --steps 20 --warmup 3 --rotate 2 2>/dev/null)
    python3 -c "
import json,sys
d=json.loads(sys.argv[1]); print('$label', '$cfg', 'step %.4f ms' % d['ms_per_step'], 'kernels', [round(k['ms'],4) for k in d['step_kernels']])" "$out"
  done
done
done
