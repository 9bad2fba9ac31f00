#!/bin/bash
# builds ntt_tree.o in the variants given as arguments ("-DSIPP_TREE_LAZY=2", "-DGLL_T=80", ...) and times each (GPU box; hipcc is there too)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R/sipp_amd/csrc
for var in "$@"; do
  echo "=== variant: $var"
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=off $var -c ntt_tree.hip -o ntt_tree.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libsipp_hip.so *.o || exit 1
  for cfg in "18 1024" "21 128"; do
    python3 $R/scripts/perf_generic.py $cfg 2>/dev/null | grep "ntt_tree\|commit" || exit 1
  done
done
