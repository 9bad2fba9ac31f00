#!/bin/bash
# A/B of ONE translation unit's flags on the WHOLE instance (GPU box): ab_bench.sh <file.hip> "<flags A>" "<flags B>" ...   (two alternating passes)
R=${GRAFT_REPO_ROOT:-$(pwd)}
SRC=$1; shift
CMD='SIPP_BENCH_IO_SHARD_N= python3 bench.py --no-cpu-baseline --steps ${STEPS:-10} --warmup 2 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(\"single %.2f ms  queue %.2f ms  pair_serial %.2f  leaves_serial %.2f\" % (d[\"ms_per_step\"], d[\"pipelined\"][\"ms_per_instance\"], d[\"kernel_ms_serial\"].get(\"poseidon_leaves_pair\",0), d[\"kernel_ms_serial\"].get(\"poseidon_leaves\",0)))"'
for pass in $(seq ${PASSES:-2}); do
  bash $R/scripts/ab_obj.sh $SRC "$CMD" "$@" || exit 1
done
