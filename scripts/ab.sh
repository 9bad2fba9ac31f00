#!/bin/bash
# usage: ab.sh "ENV1=a ENV2=b" ... : the default bench once per environment string (quoted), concurrent and serial
for e in "$@"; do
  for mode in 0 1; do
  env $e SIPP_BENCH_SERIAL=$mode python bench.py --steps 10 --warmup 2 --no-cpu-baseline --inflight 1 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); k=r['kernel_ms_per_step']
top=sorted(k.items(), key=lambda kv:-kv[1])[:9]
print('[$e] serial=$mode ms_per_step %.2f ' % r['ms_per_step'] + ' '.join('%s %.1f' % (a.replace('poseidon_','p_').replace('lookup_','lk_'),b) for a,b in top))"
  done
done
