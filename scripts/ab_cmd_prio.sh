for pass in 1 2; do
for pr in "low,,high" "low,high,high" ",high,high" "low,high,"; do
  SIPP_BENCH_PRIOS="$pr" SIPP_BENCH_IO_SHARD_N=0 SIPP_BENCH_MAP_G2=0 SIPP_BENCH_OTHER_AIR=0 SIPP_BENCH_OUTER_PLONK=0 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('[$pr] single %.2f queued %.2f' % (r['ms_per_step'], r['pipelined']['ms_per_instance']))"
done; done
