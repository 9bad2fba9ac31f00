# one variant's figures for A/B runs of the trace fill's row kernel (scripts/ab_prebuilt.sh): trace parity, then the serial event time of curve_rows per instance
python -m pytest tests/test_gpu_trace.py tests/test_gpu_hardened.py -q -x 2>&1 | tail -1
for n in 128 1024 4096; do
SIPP_BENCH_IO_SHARD_N= SIPP_BENCH_MAP_G2=0 SIPP_BENCH_OTHER_AIR=0 SIPP_BENCH_OUTER_PLONK=0 python3 bench.py --n $n --no-cpu-baseline --steps 2 --warmup 1 --inflight 1 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms_serial']
print('n=$n', 'ms_per_step %.2f'%d['ms_per_step'], 'curve_rows serial', k.get('trace_curve_rows'), 'verified', d['verified'])"
done
