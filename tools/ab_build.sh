#!/bin/bash
# developer tool: build experiment variants of librtgo_hip.so into tools/_diag/ab_<name>.so:  tools/ab_build.sh name "-DFLAG ..." [name flags ...]
set -e
cd "$(dirname "$0")/.."
while [ $# -ge 2 ]; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-slp-vectorize -mllvm -amdgpu-atomic-optimizer-strategy=None $2 \
      -o tools/_diag/ab_$1.so raytracingo_amd/csrc/rtgo_capi.hip 2>&1 | grep -i "error" || true
  echo "built ab_$1 ($2)"
  shift 2
done
