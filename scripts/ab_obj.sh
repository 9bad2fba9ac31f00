#!/bin/bash
# A/B of ONE translation unit built with different flags (GPU box; hipcc is there): ab_obj.sh <file.hip> "<cmd>" "<flags A>" "<flags B>" ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
SRC=$1; CMD=$2; shift 2
cd $R/sipp_amd/csrc
for var in "$@"; do
  echo "=== $SRC variant: [$var]"
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form -I . -I $R/data -I $R/scripts/ubench $var -c $SRC -o ${SRC%.hip}.o 2>&1 | grep -E "error|spill" 
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libsipp_hip.so *.o || exit 1
  (cd $R && eval "$CMD") || exit 1
done
