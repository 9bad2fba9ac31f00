#!/bin/bash
# A/B of the leaf-hash kernel's dense constant products: lazy multiply-adds on the VALU (round 3) against int8 products on the matrix
# pipe (poseidon.hpp::dense_mfma).  GPU box: scripts/ab_dense.sh [passes]
R=${GRAFT_REPO_ROOT:-$(pwd)}
export SIPP_BENCH_IO_SHARD_N= SIPP_BENCH_MAP_G2=0 SIPP_BENCH_OTHER_AIR=0
CMD='python3 scripts/perf_generic.py 17 1024 | grep -E "leaf perms|poseidon_leaves "; python3 scripts/perf_generic.py 21 128 | grep -E "leaf perms"; python3 bench.py --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(\"single %.2f ms  queue %.2f ms  leaves_serial %.2f\" % (d[\"ms_per_step\"], d[\"pipelined\"][\"ms_per_instance\"], d[\"kernel_ms_serial\"].get(\"poseidon_leaves\",0)))"'
for pass in $(seq ${1:-2}); do
  bash $R/scripts/ab_obj.sh poseidon.hip "$CMD" "-DSIPP_POSEIDON_VALU_DENSE" "" || exit 1
done
