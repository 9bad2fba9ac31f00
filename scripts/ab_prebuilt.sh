#!/bin/bash
# A/B of translation units PREBUILT in the container (hipcc cross-compiles there; no GPU minutes spent compiling):
#   ab_objs/<variant>/<unit>.o replace sipp_amd/csrc/<unit>.o, the library is relinked on the GPU box and "<cmd>" runs once per variant.
# usage (GPU box): scripts/ab_prebuilt.sh "<cmd>" <passes> <variant> [<variant> ...]
R=${GRAFT_REPO_ROOT:-$(pwd)}
CMD=$1; PASSES=$2; shift 2
cd $R/sipp_amd/csrc
mkdir -p /tmp/ab_base && cp *.o /tmp/ab_base/
for pass in $(seq $PASSES); do
  for var in "$@"; do
    echo "=== variant: [$var]"
    cp /tmp/ab_base/*.o . && cp $R/ab_objs/$var/*.o . || exit 1
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libsipp_hip.so *.o || exit 1
    (cd $R && eval "$CMD") || exit 1
  done
done
cp /tmp/ab_base/*.o . && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libsipp_hip.so *.o
