#!/bin/bash
# In-session A/B support: build easyanimate_amd/lib/variants/libea_old.so with ONE translation unit (default
# ea_attention.hip, with its .inc files) taken from a git ref (default HEAD) and everything else from the working tree.
#   bash tools/ab_build_old.sh [ref] [unit.hip]
# Then on the GPU box, in ONE gpurun call:  python tools/microbench.py attn ; EA_LIB_PATH=.../libea_old.so python tools/microbench.py attn
set -e
cd "$(dirname "$0")/.."
REF=${1:-HEAD}
UNIT=${2:-ea_attention.hip}
python -m easyanimate_amd.build > /dev/null
T=$(mktemp -d)
for f in $UNIT $(git ls-tree --name-only $REF easyanimate_amd/csrc/ | xargs -n1 basename | grep '\.inc$'); do git show $REF:easyanimate_amd/csrc/$f > $T/$f; done
sed "s#\"../../include/ea_mi355x.h\"#\"$PWD/include/ea_mi355x.h\"#" easyanimate_amd/csrc/ea_common.h > $T/ea_common.h
mkdir -p easyanimate_amd/lib/variants
EXTRA=""
[ "$UNIT" = ea_attention.hip ] && EXTRA="-mllvm -amdgpu-mfma-vgpr-form=1 -fno-slp-vectorize"
O=${UNIT%.hip}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value $EXTRA -x hip -c $T/$UNIT -o $T/${O}_old.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o easyanimate_amd/lib/variants/libea_old.so $(ls easyanimate_amd/build/*.o | grep -v "/$O.o") $T/${O}_old.o
rm -rf $T
echo easyanimate_amd/lib/variants/libea_old.so
