#!/bin/bash
# Round 6, second GPU session: the new tests (tiling, text encoder, LayerNorm sweep, bench robustness), the smoke canaries, a finer
# LayerNorm grid A/B, the attention ablation variants (tools/build_variants.sh "A<bits> -DEA_ATT3_ABL=<bits>").
mkdir -p gpurun_out/r6b
O=gpurun_out/r6b
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_text_encoder_gpu.py tests/test_vae_gpu.py tests/test_kernels_gpu.py -m gpu -x -q -s -k "text_encoder or attention_causal or rope_half or encode_prompt_through or tile_blend or vae_tiling or layernorm" > $O/new_tests.log 2>&1; echo "rc $?" >> $O/new_tests.log
tail -4 $O/new_tests.log
timeout 1200 python -m pytest tests/test_sequence_parallel_gpu.py -m gpu -x -q -s -k "bench" > $O/bench_tests.log 2>&1; echo "rc $?" >> $O/bench_tests.log
tail -4 $O/bench_tests.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?" >> $O/smoke.log
tail -5 $O/smoke.log
timeout 600 python tools/ab_layernorm.py 3 > $O/ab_layernorm.jsonl 2>&1
for rep in 1 2; do
  for v in cur A1 A2 A4 A8 A16 A17 A14 A31; do
    if [ "$v" = cur ]; then unset EA_LIB_PATH; else export EA_LIB_PATH=$PWD/easyanimate_amd/lib/variants/libea_$v.so; fi
    r=$(timeout 300 python tools/ab_attn_lib.py 2>&1 | grep 'attention v3' | sed 's/.*"ms": \([0-9.]*\).*/\1/' | tr '\n' ' ')
    echo "variant=$v ms: $r" >> $O/attention_ablations.txt
  done
done
unset EA_LIB_PATH
cat $O/attention_ablations.txt
