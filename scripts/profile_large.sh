#!/bin/bash
# rocprofv3 artefacts of the LARGE single-GPU configs (BASELINE configs[2] / the one-GPU leg of configs[4]):
#   profile_large.sh <tag> <n> [steps]   ->  gpurun_out/prof_<tag>_n<n>/{stats,pmc_f,pmc_w}
# kernel-trace + stats, then separate PMC passes (never combined with other trace domains); the program itself after `--`.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; N=$2; STEPS=${3:-2}
OUT=$R/gpurun_out/prof_${TAG}_n$N
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export SIPP_BENCH_IO_SHARD_N=0
export SIPP_BENCH_MAP_G2=0       # the profiled command is the instance alone: no messages -> G2 leg,
export SIPP_BENCH_OUTER_PLONK=0  # no outer-prover leg,
export SIPP_BENCH_OTHER_AIR=0    # no leg for the other AIR variant (SIPP_BENCH_PLAIN_AIR=1 in the environment profiles the plain kinds)
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o run -- python3 "$R/bench.py" --n $N --steps $STEPS --warmup 1 --no-cpu-baseline --inflight 1 > "$OUT/bench_line.json" 2> "$OUT/stats.log" || exit 1
echo "stats done $N"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_f" -o run -- python3 "$R/bench.py" --n $N --steps 1 --warmup 1 --no-cpu-baseline --inflight 1 > /dev/null 2> "$OUT/pmc_f.log" || exit 1
echo "fetch done $N"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_w" -o run -- python3 "$R/bench.py" --n $N --steps 1 --warmup 1 --no-cpu-baseline --inflight 1 > /dev/null 2> "$OUT/pmc_w.log" || exit 1
echo "write done $N"
rm -f "$OUT"/*/run_agent_info.csv
ls -la "$OUT"/*
