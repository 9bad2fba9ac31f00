for cfg in "16 1024" "18 1024" "21 128"; do python scripts/perf_generic.py $cfg 2>&1 | grep -E "commit|ntt_tree" | tr '\n' ' '; echo; done
