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
export SIPP_BENCH_PAIRING=0       # ... and without the final-pairing leg (profiled on its own at the end of this script)
export SIPP_BENCH_OUTER_PLONK=0   # ... and without the outer-prover leg (profiled on its own at the end of this script)
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o run -- python3 "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --inflight 1 > "$OUT/bench_line.json" 2> "$OUT/stats.log"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_f" -o run -- python3 "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --inflight 1 > /dev/null 2> "$OUT/pmc_f.log"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_w" -o run -- python3 "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --inflight 1 > /dev/null 2> "$OUT/pmc_w.log"
# VALU pass (round 5): instructions, the quad-cycles the VALU was busy with them (SQ_ACTIVE_INST_VALU; SQ_ACTIVE_INST_VALU2 = quad-cycles
# in which TWO instructions issued -- gfx950 pairs plain 32-bit VOP1 / VOP2 forms), and GRBM_GUI_ACTIVE for the clock the chip held
# (MI355X_MICROARCH.md, DVFS give-back: GRBM_GUI_ACTIVE / 8 / kernel wall time).  Dispatches are serialised under counter collection,
# so these are the figures of every kernel ALONE on the chip.
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VALU2 SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_v" -o run -- python3 "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --inflight 1 > /dev/null 2> "$OUT/pmc_v.log"
# the same counters over the instruction-form micro-benchmark: what the counters MEAN (a stream of v_add_u32 pairs: half a quad-cycle
# per instruction; carry / 64-bit / multiply forms: one quad-cycle each) -- the calibration bench.py's `valu` object rests on.
# The binaries are git-ignored: built HERE, and every calibration step must succeed -- an empty enc_rates.txt next to a fresh
# source_sha256.txt would stamp counters as current whose calibration is missing (scripts/summarize_profile.py refuses that too).
mkdir -p "$R/scripts/ubench/bin"
for b in enc_rates canon_rates; do
    if [ ! -x "$R/scripts/ubench/bin/$b" ] || [ "$R/scripts/ubench/$b.hip" -nt "$R/scripts/ubench/bin/$b" ]; then
        /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -I "$R/sipp_amd/csrc" -o "$R/scripts/ubench/bin/$b" "$R/scripts/ubench/$b.hip" || { echo "profile_round: cannot build scripts/ubench/$b" >&2; exit 1; }
    fi
done
"$R/scripts/ubench/bin/enc_rates" > "$OUT/enc_rates.txt" 2> "$OUT/enc_rates.log" || { echo "profile_round: enc_rates failed" >&2; exit 1; }
"$R/scripts/ubench/bin/canon_rates" > "$OUT/canon_rates.txt" 2>> "$OUT/enc_rates.log" || { echo "profile_round: canon_rates failed" >&2; exit 1; }
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VALU2 SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_enc" -o run -- "$R/scripts/ubench/bin/enc_rates" > "$OUT/pmc_enc.txt" 2> "$OUT/pmc_enc.log" || { echo "profile_round: the counter pass over enc_rates failed" >&2; exit 1; }
[ -s "$OUT/enc_rates.txt" ] || { echo "profile_round: empty calibration output" >&2; exit 1; }
# which code the counters belong to (bench.py compares it with the tree it runs from)
python3 "$R/sipp_amd/build.py" hash > "$OUT/source_sha256.txt"
# concurrency timeline of the same command (kernel trace only)
rocprofv3 --kernel-trace --output-format csv -d "$OUT/tl" -o run -- python3 "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --inflight 1 > /dev/null 2> "$OUT/tl.log"
# native chain
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/native" -o run -- python3 "$R/scripts/perf_native.py" 128 > "$OUT/native.txt" 2> "$OUT/native.log"
# the messages -> G2 step of the BLS example (HISTORY.md section 7b)
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/mapg2" -o run -- python3 "$R/scripts/perf_mapg2.py" 127 > "$OUT/mapg2.txt" 2> "$OUT/mapg2.log"
# the hardened G1 / G2 AIRs beside the plain ones
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/hardened" -o run -- python3 "$R/scripts/perf_hardened.py" > "$OUT/hardened.txt" 2> "$OUT/hardened.log"
# the outer plonky2 prover at the reference's circuit configuration, gates as data (bench.py `outer_plonk`)
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/plonk" -o run -- python3 "$R/scripts/perf_plonk.py" 18 5 > "$OUT/plonk.txt" 2> "$OUT/plonk.log"
# the final-pairing STARK of the BLS example (kind 6, round 6): one record
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/pairing" -o run -- python3 "$R/scripts/perf_pairing_stark.py" 1 5 > "$OUT/pairing.txt" 2> "$OUT/pairing.log"
ls -R "$OUT" | head -40
