# transform A/B at the 2^21-row shape (n = 4096): parity of the commitment at 2^21, then the tree kernels
python -m pytest tests/test_gpu_generic.py -q -x -k "commit_matches_oracle and (21-1 or 18-2) or long_column" 2>&1 | tail -1
for cfg in "21 128" "21 256"; do python scripts/perf_generic.py $cfg 2>&1 | grep -E "commit|ntt_tree" | tr '\n' ' '; echo; done
