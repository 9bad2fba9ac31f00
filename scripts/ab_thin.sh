#!/bin/bash
# A/B of the thin trees' leaf kernel (the Fq12 STARK's 2^14-leaf trees: the lone instance's long pole): two lanes per state (default),
# the same with the hand-scheduled Goldilocks product, four lanes per state.  GPU box: scripts/ab_thin.sh [passes]
R=${GRAFT_REPO_ROOT:-$(pwd)}
export SIPP_BENCH_IO_SHARD_N= SIPP_BENCH_MAP_G2=0 SIPP_BENCH_OTHER_AIR=0
CMD='python3 bench.py --no-cpu-baseline --steps 15 --warmup 3 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); ks=d[\"kernel_ms_serial\"]; print(\"single %.2f ms  queue %.2f ms  thin_serial %.2f\" % (d[\"ms_per_step\"], d[\"pipelined\"][\"ms_per_instance\"], ks.get(\"poseidon_leaves_pair\",0)+ks.get(\"poseidon_leaves_quad\",0)))"'
for pass in $(seq ${1:-2}); do
  bash $R/scripts/ab_obj.sh poseidon.hip "$CMD" "-DSIPP_THIN_LANES=4" "-DSIPP_POSEIDON_THIN_ASM_MUL" "" || exit 1
done
