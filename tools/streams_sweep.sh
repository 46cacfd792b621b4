#!/bin/bash
# Run ON THE GPU BOX: step time of one config over batches in flight.  tools/streams_sweep.sh c3 "1 2 3"
cfg=$1; reads=1000000; [ $cfg = c2 ] && reads=765000
for rep in 1 2; do for st in $2; do
  out=$(python3 bench.py --no-cpu-baseline --no-extras --config $cfg --reads $reads --steps 20 --warmup 3 --rotate 2 --streams $st 2>/dev/null)
  python3 -c "
import json,sys
d=json.loads(sys.argv[1]); print('$cfg streams=$st', 'step %.4f ms' % d['ms_per_step'], '%.3e reads/s' % d['value'])" "$out"
done; done
