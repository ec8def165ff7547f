#!/bin/bash
# In-session A/B support: build easyanimate_amd/lib/variants/libea_old.so with the named translation units (default
# ea_attention.hip; .inc files always come along) taken from a git ref (default HEAD) and everything else from the
# working tree.     bash tools/ab_build_old.sh [ref] [unit ...]
# Then on the GPU box, in ONE gpurun call:  bash tools/ab_attn.sh cur old cur old
set -e
cd "$(dirname "$0")/.."
REF=${1:-HEAD}
shift || true
UNITS=${@:-ea_attention.hip}
python -m easyanimate_amd.build > /dev/null
T=$(mktemp -d)
for f in $UNITS $(git ls-tree --name-only $REF easyanimate_amd/csrc/ | xargs -n1 basename | grep '\.inc$'); do git show $REF:easyanimate_amd/csrc/$f > $T/$f; done
sed "s#\"../../include/ea_mi355x.h\"#\"$PWD/include/ea_mi355x.h\"#" easyanimate_amd/csrc/ea_common.h > $T/ea_common.h
mkdir -p easyanimate_amd/lib/variants
SKIP=""
OBJS=""
for u in $UNITS; do
  O=${u%.*}
  EXTRA=""
  [ "$u" = ea_attention.hip ] && EXTRA="-mllvm -amdgpu-mfma-vgpr-form=1 -fno-slp-vectorize"
  X="-x hip"; [ "${u##*.}" = cpp ] && X=""
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value $EXTRA $X -c $T/$u -o $T/${O}_old.o
  SKIP="$SKIP -e /$O.o"
  OBJS="$OBJS $T/${O}_old.o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o easyanimate_amd/lib/variants/libea_old.so $(ls easyanimate_amd/build/*.o | grep -v $SKIP) $OBJS
rm -rf $T
echo easyanimate_amd/lib/variants/libea_old.so
