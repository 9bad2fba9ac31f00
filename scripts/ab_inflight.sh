#!/bin/bash
# queue depth sweep of bench.py's `pipelined` figure (GPU box): ab_inflight.sh 5 6 8 ...   (alternating passes)
export SIPP_BENCH_IO_SHARD_N= SIPP_BENCH_MAP_G2=0 SIPP_BENCH_OTHER_AIR=0
for pass in $(seq ${PASSES:-2}); do
  for k in "$@"; do
    echo -n "[inflight $k] "
    python3 bench.py --no-cpu-baseline --steps 10 --warmup 2 --inflight $k 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('single %.2f ms  queue %.2f ms' % (d['ms_per_step'], d['pipelined']['ms_per_instance']))" || exit 1
  done
done
