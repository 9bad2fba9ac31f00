# transform A/B at 2^17 / 2^18 rows: parity, then the tree kernels
python -m pytest tests/test_gpu_generic.py -q -x -k "commit_matches_oracle or long_column" 2>&1 | tail -1
for cfg in "17 1024" "18 1024"; do python scripts/perf_generic.py $cfg 2>&1 | grep -E "commit|ntt_tree" | tr '\n' ' '; echo; done
