#!/bin/bash
# Build libea variants that differ in the flags of ea_attention.hip only (compiler-scheduling A/B):
#   bash tools/build_variants.sh   -> easyanimate_amd/lib/variants/libea_<tag>.so ; run with EA_LIB_PATH=<that file>
set -e
cd "$(dirname "$0")/.."
python -m easyanimate_amd.build > /dev/null
OUT=easyanimate_amd/lib/variants
mkdir -p $OUT
OBJ=easyanimate_amd/build
COMMON="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -mllvm -amdgpu-mfma-vgpr-form=1"
build() {  # tag, extra flags...
  tag=$1; shift
  /opt/rocm/bin/hipcc $COMMON "$@" -x hip -c easyanimate_amd/csrc/ea_attention.hip -o $OUT/ea_attention_$tag.o
  objs=$(ls $OBJ/*.o | grep -v ea_attention.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libea_$tag.so $objs $OUT/ea_attention_$tag.o
  rm $OUT/ea_attention_$tag.o
  echo built $tag
}
# (measured in round 1: F 1141 TF > D 1133 > E 1112 > A 1105; -amdgpu-igrouplp-exact-solver needs
#  -amdgpu-igrouplp-exact-solver-max-branches=<N> and a `timeout`: uncapped it ran for > 30 minutes on this kernel)
COMMON="$COMMON -fno-slp-vectorize"
build MS0 -DEA_ATT3_MFMASUM=0 &
wait
ls $OUT
