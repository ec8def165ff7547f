#!/bin/bash
# In-session A/B support: build easyanimate_amd/lib/variants/libea_old.so with ea_attention*.{hip,inc} taken from a git
# ref (default HEAD) and everything else from the working tree.   bash tools/ab_build_old.sh [ref]
# Then on the GPU box, in ONE gpurun call:  python tools/microbench.py attn ; EA_LIB_PATH=.../libea_old.so python tools/microbench.py attn
set -e
cd "$(dirname "$0")/.."
REF=${1:-HEAD}
python -m easyanimate_amd.build > /dev/null
T=$(mktemp -d)
for f in ea_attention.hip ea_attention_v2.inc ea_attention_v3.inc; do git show $REF:easyanimate_amd/csrc/$f > $T/$f; done
sed "s#\"../../include/ea_mi355x.h\"#\"$PWD/include/ea_mi355x.h\"#" easyanimate_amd/csrc/ea_common.h > $T/ea_common.h
mkdir -p easyanimate_amd/lib/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -mllvm -amdgpu-mfma-vgpr-form=1 -x hip -c $T/ea_attention.hip -o $T/ea_attention_old.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o easyanimate_amd/lib/variants/libea_old.so $(ls easyanimate_amd/build/*.o | grep -v ea_attention.o) $T/ea_attention_old.o
rm -rf $T
echo easyanimate_amd/lib/variants/libea_old.so
