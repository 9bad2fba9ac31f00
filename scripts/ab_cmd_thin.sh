# one variant's figures for the leaf-hash A/B runs (scripts/ab_prebuilt.sh): parity of the hash kernels, then the thin (2^14 x 4942) and
# fat (2^18 x 1024) leaf kernels alone, then the n = 128 instance single / queued
python -m pytest tests/test_gpu_generic.py -q -x -k "poseidon or commit_matches_oracle or leaves" 2>&1 | tail -1
python scripts/perf_generic.py 13 4942 2>&1 | grep -E "perms"
python scripts/perf_generic.py 17 1024 2>&1 | grep -E "perms"
SIPP_BENCH_IO_SHARD_N= SIPP_BENCH_MAP_G2=0 SIPP_BENCH_OTHER_AIR=0 python3 bench.py --no-cpu-baseline --steps 15 --warmup 3 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); ks=d['kernel_ms_serial']; print('single %.2f ms  queue %.2f ms  thin_serial %.2f  fat_serial %.2f' % (d['ms_per_step'], d['pipelined']['ms_per_instance'], ks.get('poseidon_leaves_pair',0), ks.get('poseidon_leaves',0)))"
