#!/bin/bash
# A/B of environment settings on the WHOLE instance (GPU box): ab_env.sh "<VAR=..  VAR=..>" "<...>" ...   ("-" = no setting; alternating passes)
CMD='python3 bench.py --no-cpu-baseline --steps ${STEPS:-10} --warmup 2 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(\"single %.2f ms  queue %.2f ms  pair_serial %.2f  leaves_serial %.2f\" % (d[\"ms_per_step\"], d[\"pipelined\"][\"ms_per_instance\"], d[\"kernel_ms_serial\"].get(\"poseidon_leaves_pair\",0), d[\"kernel_ms_serial\"].get(\"poseidon_leaves\",0)))"'
export SIPP_BENCH_IO_SHARD_N= SIPP_BENCH_MAP_G2=0 SIPP_BENCH_OTHER_AIR=0
for pass in $(seq ${PASSES:-2}); do
  for setting in "$@"; do
    echo -n "[$setting] "
    if [ "$setting" = "-" ]; then bash -c "$CMD" || exit 1; else bash -c "export $setting; $CMD" || exit 1; fi
  done
done
