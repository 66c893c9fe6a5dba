#!/bin/bash
# developer tool: VGPR / scratch / occupancy of every kernel of librtgo_hip.so (hipcc remarks), one line per kernel
cd /tmp && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-slp-vectorize -mllvm -amdgpu-atomic-optimizer-strategy=None "$@" \
  -Rpass-analysis=kernel-resource-usage -o /tmp/rtgo_ru.so /root/repo/raytracingo_amd/csrc/rtgo_capi.hip 2>&1 |
  grep -E "Function Name|VGPRs:|ScratchSize|Occupancy" | sed -E 's/.*remark: +//; s/ \[-Rpass.*//; s/Function Name: _ZN4rtgo[0-9]*/@/; s/EvNS_12LaunchParams.*//; s/EPKNS_6PrimIn.*//' |
  tr '\n' ' ' | tr '@' '\n'; echo
