#!/bin/bash
# build_variant.sh NAME "<extra hipcc flags>" [file.hip]: a second libquokka_amd with one translation unit compiled with extra -D flags
# (quokka_amd/lib/variants/libqk_NAME.so; select it with QK_LIB_PATH) for same-box A/B runs
set -e
name=$1; flags=$2; unit=${3:-qk_hydro_fused.hip}
cd "$(dirname "$0")/../../quokka_amd/csrc"
mkdir -p ../lib/variants
/opt/rocm/bin/hipcc $flags -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -I../../include -I. -Wall -Wno-unused-function -c $unit -o /tmp/variant_$name.o
objs=$(ls *.o | grep -v "^${unit%.hip}.o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/variants/libqk_$name.so /tmp/variant_$name.o $objs
echo built ../lib/variants/libqk_$name.so
