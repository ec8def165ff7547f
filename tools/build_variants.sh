#!/bin/bash
# Build libea variants that differ in the flags of ONE source file only (compiler-scheduling / experiment-switch A/B), on top of
# the EA_BUILD_VARIANTS=1 library (so every cross-check / experiment kernel is selectable in them):
#   bash tools/build_variants.sh [-f ea_gemm.hip] "TAG -DEA_ATT4_FINE=2" "OTHER -DEA_ATT3_LEAD=1 ..."
#       -> easyanimate_amd/lib/variants/libea_<TAG>.so (run with EA_LIB_PATH=<that file>; tools/ab_attn_lib.py, tools/ab_gemm.py)
#          and the device assembly in /tmp/<file>_<TAG>.s (instruction mix of the hot block: tools/isa_hot_block.py)
set -e
cd "$(dirname "$0")/.."
SRC=ea_attention.hip
if [ "$1" = "-f" ]; then SRC=$2; shift 2; fi
BASE=${SRC%.hip}
OUT=easyanimate_amd/lib/variants
mkdir -p $OUT
EA_BUILD_VARIANTS=1 EA_LIB_OUT=$OUT/libea_variants.so python -m easyanimate_amd.build > /dev/null
OBJ=easyanimate_amd/build/out_libea_variants.so
# (measured in round 1: -amdgpu-igrouplp-exact-solver needs -amdgpu-igrouplp-exact-solver-max-branches=<N> and a `timeout`:
#  uncapped it ran for > 30 minutes on the attention kernel)
COMMON="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -DEA_BUILD_VARIANTS=1"
if [ $SRC = ea_attention.hip ]; then COMMON="$COMMON -mllvm -amdgpu-mfma-vgpr-form=1 -fno-slp-vectorize"; fi
build() {  # tag, extra flags...
  tag=$1; shift
  /opt/rocm/bin/hipcc $COMMON "$@" -x hip -c easyanimate_amd/csrc/$SRC -o $OUT/${BASE}_$tag.o 2> /dev/null
  objs=$(ls $OBJ/*.o | grep -v "/$BASE.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libea_$tag.so $objs $OUT/${BASE}_$tag.o
  rm $OUT/${BASE}_$tag.o
  /opt/rocm/bin/hipcc $COMMON "$@" --cuda-device-only -S -x hip easyanimate_amd/csrc/$SRC -o /tmp/${BASE}_$tag.s 2> /dev/null
  echo built $tag
}
if [ $# -eq 0 ]; then set -- "MS0 -DEA_ATT3_MFMASUM=0"; fi
for spec in "$@"; do
  build $spec &
done
wait
ls $OUT
