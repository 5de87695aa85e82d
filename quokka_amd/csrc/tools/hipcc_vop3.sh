#!/bin/bash
# hipcc_vop3.sh <in.hip> <out.o> <flags...> : compile one HIP translation unit for gfx950 with every VOP2 v_cndmask_b32 of the device code
# re-encoded as VOP3 (tools/vop3_cndmask.py says why).  Same steps as `hipcc -c` (hipcc -### shows them): device code -> assembly
# [-> re-encode] -> object -> code object -> offload bundle -> host object embedding the bundle.
set -e
IN=$1; OUT=$2; shift 2
LLVM=/opt/rocm/lib/llvm/bin
T=${OUT%.o}
HERE=$(dirname "$0")
/opt/rocm/bin/hipcc "$@" --cuda-device-only -S "$IN" -o "$T.dev.s"
python3 "$HERE/vop3_cndmask.py" "$T.dev.s" "$T.dev.vop3.s"
$LLVM/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c "$T.dev.vop3.s" -o "$T.dev.o"
$LLVM/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -o "$T.hsaco" "$T.dev.o"
$LLVM/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 -input=/dev/null -input="$T.hsaco" -output="$T.hipfb"
/opt/rocm/bin/hipcc "$@" --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang "$T.hipfb" -c "$IN" -o "$OUT"
rm -f "$T.dev.s" "$T.dev.o" "$T.hsaco" "$T.hipfb"
