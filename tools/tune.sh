#!/bin/bash
# quick A/B of tile size / occupancy knobs on the GPU box: tools/tune.sh "R=64" "R=32 B=4" ...
cd "$GRAFT_REPO_ROOT"
for cfg in "$@"; do
  env_args=""
  for kv in $cfg; do
    case $kv in
      R=*) env_args="$env_args SMX_TILE_R=${kv#R=}";;
      B=*) env_args="$env_args SMX_BLOCKS_PER_CU=${kv#B=}";;
      L=*) env_args="$env_args SMX_LDS_BUDGET=${kv#L=}";;
      T=*) env_args="$env_args SMX_PHASE_TIMING=1";;
      P=*) env_args="$env_args SMX_LDS_PAD=${kv#P=}";;
      D=*) env_args="$env_args SMX_DEBUG=1";;
      X=*) env_args="$env_args SMX_LIB=$GRAFT_REPO_ROOT/build_exp/libsmx_${kv#X=}.so";;
    esac
  done
  out=$(env $env_args python bench.py --no-cpu-baseline --steps 10 --warmup 2 2>gpurun_out/tune.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.1f Mreads/s kernel %.3f ms' % (d['value']/1e6, d['roofline']['kernel_ms_avg']))")
  echo "[$cfg] $out $(grep -E "phase timing|occupancy API|placement" gpurun_out/tune.err | tail -20 | tr '\n' ' ')"
done
