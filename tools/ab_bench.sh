#!/bin/bash
# Run ON THE GPU BOX: kernel-resident bench (no CPU baseline, no extras) for c2 / c3 / c5 under a list of environments.
#   tools/ab_bench.sh "SMX_COMPACT=0" "SMX_NO_SPECIALISE=1" ...   -> one line per (env, config): reads/s, step ms, kernel ms
for envs in "$@"; do
  for cfg in c2 c3 c5; do
    reads=765000; [ $cfg != c2 ] && reads=1000000
    out=$(env $envs python3 bench.py --no-cpu-baseline --no-extras --config $cfg --reads $reads --steps 10 --warmup 3 --rotate 2 2>/dev/null)
    python3 -c "
import json,sys
d=json.loads(sys.argv[1]); print('$envs', '$cfg', '%.3e reads/s' % d['value'], 'step %.4f ms' % d['ms_per_step'], 'kernels', [round(k['ms'],4) for k in d['step_kernels']])" "$out"
  done
done
