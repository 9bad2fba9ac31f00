#!/bin/bash
# A/B of the tree-of-rings transforms (ntt_tree.hip) against the other paths (GPU box): SIPP_TREE_MIN_LOG=30 switches them off
R=${GRAFT_REPO_ROOT:-$(pwd)}
for cfg in "$@"; do
  for mode in "SIPP_TREE_MIN_LOG=30" "SIPP_TREE_MIN_LOG=15"; do
    echo "== $cfg $mode"
    env $mode python3 $R/scripts/perf_generic.py $cfg 2>/dev/null | grep -v "leaf perms\|lde bytes\|merkle\|poseidon" || exit 1
  done
done
