#!/bin/bash
# usage: tools/slice_regs.sh <slice 0..3> <shape id> [extra -D flags]: compile ONE shape slice and print the VGPRs of its k_round / k_tail kernels
set -e
SL=$1; SH=$2; shift 2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result -pthread -DSVSDF_SLICE=$SL "$@" -c $ROOT/implicit-svsdf-planner_amd/csrc/svsdf_shape_slice.hip -o $T/s.o 2>&1 | grep -E "error" || true
/opt/rocm/lib/llvm/bin/llvm-objcopy --dump-section .hip_fatbin=$T/fat.bin $T/s.o $T/copy.o
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=$T/fat.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$T/dev.co
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $T/dev.co 2>/dev/null | grep -E "^\s+\.name:|\.vgpr_count|vgpr_spill" | paste - - - | grep -E "k_round|k_tail" | grep "Li${SH}E" | sed -E 's/.*(k_round|k_tail)ILi([0-9]+)ELi([0-9]+)E(Li([0-9])E)?.*vgpr_count: *([0-9]+).*spill_count: *([0-9]+)/\1<\2,\3,\5> vgpr \6 spill \7/' | sort
rm -rf $T
