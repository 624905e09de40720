#!/bin/bash
# scripts/kernel_resources.sh [extra hipcc flags] — VGPR / SGPR / scratch / LDS / occupancy of the kernels (compiler remarks)
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc -Os --offload-arch=gfx950 -std=c++17 -Iinclude -Ipomcpp_amd/csrc --cuda-device-only -c -o /dev/null \
  -Rpass-analysis=kernel-resource-usage "$@" pomcpp_amd/csrc/pom_batch.hip 2>&1 | grep "remark:" | sed 's/ \[-Rpass.*//' | \
  awk '/Function Name:/{name=$NF} / VGPRs:/{v=$NF} /TotalSGPRs:/{s=$NF} /ScratchSize/{p=$NF} /Occupancy/{o=$NF} /VGPRs Spill/{sp=$NF} /LDS Size/{if (name ~ /pom_(step|policy)_kernel/) printf "%-62s vgpr %3s sgpr %3s scratch %4s vspill %3s occupancy %2s lds %6s\n", name, v, s, p, sp, o, $NF}'
