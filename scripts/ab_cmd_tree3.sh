# one variant's figures for transform A/B runs (scripts/ab_prebuilt.sh): parity at the tree sizes, then the tree kernels at three shapes
python -m pytest tests/test_gpu_generic.py -q -x -k "commit_matches_oracle and (15-3 or 16-5 or 18-2 or 21-1) or long_column" 2>&1 | tail -1
for cfg in "16 1024" "18 1024" "21 128"; do python scripts/perf_generic.py $cfg 2>&1 | grep -E "commit|ntt_tree" | tr '\n' ' '; echo; done
