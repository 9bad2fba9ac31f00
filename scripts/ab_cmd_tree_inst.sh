# transform A/B with the instance's figures: parity at the tree sizes, the tree kernels alone at two shapes, the n = 128 instance single / queued
python -m pytest tests/test_gpu_generic.py -q -x -k "commit_matches_oracle and (15-3 or 16-5 or 18-2 or 21-1) or long_column" 2>&1 | tail -1
for cfg in "16 1024" "21 128"; do python scripts/perf_generic.py $cfg 2>&1 | grep -E "commit|ntt_tree" | tr '\n' ' ' | cut -c1-300; echo; done
bash scripts/ab_cmd_inst.sh
