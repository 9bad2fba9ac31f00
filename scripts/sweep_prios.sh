export SIPP_BENCH_IO_SHARD_N= SIPP_BENCH_MAP_G2=0 SIPP_BENCH_OTHER_AIR=0
for pass in 1 2; do
for P in "low,high,high" "low,high," "low,,high" ",high,high" "low,high,low" ",high," "high,high,high"; do
SIPP_BENCH_PRIOS="$P" python3 bench.py --no-cpu-baseline --steps 15 --warmup 3 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$P] single %.2f ms  queue %.2f ms' % (d['ms_per_step'], d['pipelined']['ms_per_instance']))"
done; done
