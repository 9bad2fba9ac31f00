# round 6, VERDICT r5 item 4: transform A/B at the n = 1024 / n = 4096 shapes (2^19, 2^21 rows) and the n = 128 ones: parity, then the tree kernels
python -m pytest tests/test_gpu_generic.py -q -x -k "commit_matches_oracle or long_column" 2>&1 | tail -1
for cfg in "14 2048" "16 1024" "18 1024" "19 512" "21 128"; do python scripts/perf_generic.py $cfg 2>&1 | grep -E "commit|ntt_tree" | tr '\n' ' '; echo; done
