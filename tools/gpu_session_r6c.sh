#!/bin/bash
mkdir -p gpurun_out/r6c
O=gpurun_out/r6c
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_vae_gpu.py tests/test_kernels_gpu.py -m gpu -q -s -k "tile_blend or vae_tiling or layernorm" > $O/new_tests.log 2>&1; echo "rc $?" >> $O/new_tests.log
tail -4 $O/new_tests.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?" >> $O/smoke.log
tail -5 $O/smoke.log
for rep in 1 2; do
  for v in cur A1 A2 A4 A8 A16 A17 A14 A31; do
    if [ "$v" = cur ]; then unset EA_LIB_PATH; else export EA_LIB_PATH=$PWD/easyanimate_amd/lib/variants/libea_$v.so; fi
    r=$(timeout 300 python tools/ab_attn_lib.py 2>&1 | grep 'attention v3' | sed 's/.*"ms": \([0-9.]*\).*/\1/' | tr '\n' ' ')
    echo "variant=$v ms: $r" >> $O/attention_ablations.txt
  done
done
unset EA_LIB_PATH
cat $O/attention_ablations.txt
