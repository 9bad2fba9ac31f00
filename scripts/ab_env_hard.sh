#!/bin/bash
# like ab_env.sh, with the other AIR variant's leg of bench.py (recalled_upstream_air; hardened_instance under SIPP_BENCH_PLAIN_AIR=1): ab_env_hard.sh "<VAR=..>" ...   ("-" = no setting)
CMD='python3 bench.py --no-cpu-baseline --steps ${STEPS:-10} --warmup 2 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(\"headline %.2f ms  other AIR variant %.2f ms\" % (d[\"ms_per_step\"], (d.get(\"recalled_upstream_air\") or d.get(\"hardened_instance\"))[\"ms_per_instance\"]))"'
export SIPP_BENCH_IO_SHARD_N= SIPP_BENCH_MAP_G2=0 SIPP_BENCH_OTHER_AIR=1
for pass in $(seq ${PASSES:-2}); do
  for setting in "$@"; do
    echo -n "[$setting] "
    if [ "$setting" = "-" ]; then bash -c "$CMD" || exit 1; else bash -c "export $setting; $CMD" || exit 1; fi
  done
done
