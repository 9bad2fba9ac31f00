#!/bin/bash
# usage: sweep_env.sh VAR v1 v2 ... : default bench (no CPU baseline, no pipelined phase) once per value of the environment variable
VAR=$1; shift
for v in "$@"; do
  env $VAR=$v python bench.py --steps ${SWEEP_STEPS:-10} --warmup 2 --no-cpu-baseline --inflight 1 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); k=r['kernel_ms_per_step']
print('$VAR=$v  ms_per_step %.2f  leaves %.1f merkle %.1f hist %.1f scan %.1f z_a %.1f' % (r['ms_per_step'], k.get('poseidon_leaves',0), k.get('merkle_subtree',k.get('merkle_level',0)), k.get('lookup_hist',0), k.get('lookup_scan',0), k.get('z_phase_a',0)))"
done
