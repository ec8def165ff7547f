#!/bin/bash
# Build libea variants that differ in the flags of ea_attention.hip only (compiler-scheduling / experiment-switch A/B):
#   bash tools/build_variants.sh "TAG -DEA_ATT3_PPG=2" "OTHER -DEA_ATT3_LEAD=1 ..."
#       -> easyanimate_amd/lib/variants/libea_<TAG>.so (run with EA_LIB_PATH=<that file>; tools/ab_attn_lib.py) and the device
#          assembly in /tmp/att_<TAG>.s (instruction mix of the hot block: tools/isa_hot_block.py)
set -e
cd "$(dirname "$0")/.."
python -m easyanimate_amd.build > /dev/null
OUT=easyanimate_amd/lib/variants
mkdir -p $OUT
OBJ=easyanimate_amd/build
# (measured in round 1: -amdgpu-igrouplp-exact-solver needs -amdgpu-igrouplp-exact-solver-max-branches=<N> and a `timeout`:
#  uncapped it ran for > 30 minutes on this kernel)
COMMON="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -mllvm -amdgpu-mfma-vgpr-form=1 -fno-slp-vectorize"
build() {  # tag, extra flags...
  tag=$1; shift
  /opt/rocm/bin/hipcc $COMMON "$@" -x hip -c easyanimate_amd/csrc/ea_attention.hip -o $OUT/ea_attention_$tag.o 2> /dev/null
  objs=$(ls $OBJ/*.o | grep -v ea_attention.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libea_$tag.so $objs $OUT/ea_attention_$tag.o
  rm $OUT/ea_attention_$tag.o
  /opt/rocm/bin/hipcc $COMMON "$@" --cuda-device-only -S -x hip easyanimate_amd/csrc/ea_attention.hip -o /tmp/att_$tag.s 2> /dev/null
  echo built $tag
}
if [ $# -eq 0 ]; then set -- "MS0 -DEA_ATT3_MFMASUM=0"; fi
for spec in "$@"; do
  build $spec &
done
wait
ls $OUT
