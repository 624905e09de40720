#!/bin/bash
# code size of the step kernels in a built library: scripts/kernel_size.sh [lib.so]
LIB=$(realpath ${1:-pomcpp_amd/libpom_batch.so})
T=$(mktemp -d)
cd $T
/opt/rocm/lib/llvm/bin/llvm-objcopy -O binary --only-section=.hip_fatbin $LIB fat.bin
TGT=$(/opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --input=fat.bin --list | grep gfx950)
/opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --input=fat.bin --targets=$TGT --output=dev.co --unbundle
/opt/rocm/lib/llvm/bin/llvm-readelf -s --wide dev.co | awk '$4=="FUNC" {printf "%8d bytes  %s\n", $3, $8}' | sort -n
rm -rf $T
