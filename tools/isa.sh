#!/bin/bash
# Disassemble one kernel of the built shim object:  tools/isa.sh <mangled-name-substring> > out.s   (developer tool)
set -e
T=$(mktemp -d)
objcopy -O binary --only-section=.hip_fatbin "$(dirname "$0")/../spmv_amd/build/spmv_shim.hip.o" $T/fat.bin
/opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$T/fat.bin --output=$T/dev.co --unbundle
/opt/rocm/lib/llvm/bin/llvm-objdump -d --no-show-raw-insn $T/dev.co | awk -v k="$1" 'index($0, "<") && index($0, k) && /^[0-9a-f]+ </ {f=1} f {print} f && /s_endpgm/ {exit}'
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $T/dev.co | grep -B2 -A30 "$1" | grep -E "\.name:|vgpr_count|agpr_count|sgpr_count|vgpr_spill|private_segment_fixed" | head -12 >&2
rm -rf $T
