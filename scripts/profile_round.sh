#!/bin/bash
# Collects the rocprofv3 artefacts summarised under profiles/ (run on the GPU box from the repo root):
#   kernel-trace + stats of the default bench, then separate PMC passes (never combined with other trace domains).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$1
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export SIPP_BENCH_IO_SHARD_N=0   # the profiled command is the n = 128 line alone (no io_sharded leg)
export SIPP_BENCH_MAP_G2=0       # ... and without the messages -> G2 leg (profiled on its own at the end of this script)
export SIPP_BENCH_OTHER_AIR=0     # ... and without the other AIR variant's leg
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o run -- python3 "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --inflight 1 > "$OUT/bench_line.json" 2> "$OUT/stats.log"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_f" -o run -- python3 "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --inflight 1 > /dev/null 2> "$OUT/pmc_f.log"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_w" -o run -- python3 "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --inflight 1 > /dev/null 2> "$OUT/pmc_w.log"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU --output-format csv -d "$OUT/pmc_v" -o run -- python3 "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --inflight 1 > /dev/null 2> "$OUT/pmc_v.log"
# concurrency timeline of the same command (kernel trace only)
rocprofv3 --kernel-trace --output-format csv -d "$OUT/tl" -o run -- python3 "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --inflight 1 > /dev/null 2> "$OUT/tl.log"
# native chain
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/native" -o run -- python3 "$R/scripts/perf_native.py" 128 > "$OUT/native.txt" 2> "$OUT/native.log"
# the messages -> G2 step of the BLS example (HISTORY.md section 7b)
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/mapg2" -o run -- python3 "$R/scripts/perf_mapg2.py" 127 > "$OUT/mapg2.txt" 2> "$OUT/mapg2.log"
# the hardened G1 / G2 AIRs beside the plain ones
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/hardened" -o run -- python3 "$R/scripts/perf_hardened.py" > "$OUT/hardened.txt" 2> "$OUT/hardened.log"
ls -R "$OUT" | head -40
