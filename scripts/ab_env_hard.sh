#!/bin/bash
# like ab_env.sh, for the hardened-AIR leg of bench.py: ab_env_hard.sh "<VAR=..>" ...   ("-" = no setting)
CMD='python3 bench.py --no-cpu-baseline --steps ${STEPS:-10} --warmup 2 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(\"single %.2f ms  hardened %.2f ms\" % (d[\"ms_per_step\"], d[\"hardened_instance\"][\"ms_per_instance\"]))"'
export SIPP_BENCH_IO_SHARD_N= SIPP_BENCH_MAP_G2=0 SIPP_BENCH_HARDENED=1
for pass in $(seq ${PASSES:-2}); do
  for setting in "$@"; do
    echo -n "[$setting] "
    if [ "$setting" = "-" ]; then bash -c "$CMD" || exit 1; else bash -c "export $setting; $CMD" || exit 1; fi
  done
done
