#!/bin/bash
# A/B of the tree-of-rings transforms (ntt_tree.hip) against the other paths (GPU box): ab_tree.sh "<env A>" "<env B>" ... -- "<log_n cols>" ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
modes=(); while [ "$1" != "--" ]; do modes+=("$1"); shift; done; shift
for cfg in "$@"; do
  for mode in "${modes[@]}"; do
    echo "== $cfg $mode"
    env $mode python3 $R/scripts/perf_generic.py $cfg 2>/dev/null | grep -v "leaf perms\|lde bytes\|merkle\|poseidon" || exit 1
  done
done
