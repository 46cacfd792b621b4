for rep in 1 2; do
for cfg in "32 3" "32 4" "24 4" "24 5"; do set -- $cfg
  out=$(SMX_COMPACT_R=$1 SMX_BLOCKS_PER_CU=$2 python3 bench.py --no-cpu-baseline --no-extras --config c3 --reads 1000000 --steps 20 --warmup 3 --rotate 2 2>/dev/null)
  python3 -c "
import json,sys
d=json.loads(sys.argv[1]); print('Rc=$1 blocks=$2', 'step %.4f ms' % d['ms_per_step'], 'kernels', [round(k['ms'],4) for k in d['step_kernels']])" "$out"
done; done
