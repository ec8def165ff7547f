#!/bin/bash
# First GPU call of a round that starts with untested experiment kernels: correctness first, each
# step under its own `timeout` (a deadlocked experimental kernel must not run into gpurun's limit), then the in-process A/Bs.
#     bash tools/run_experiments.sh [tag]      -> gpurun_out/<tag>_*.txt     (libraries: tools/build_variants.sh, built on CPU)
TAG=${1:-r04a}
V=$PWD/easyanimate_amd/lib/variants
mkdir -p gpurun_out
export EA_LIB_PATH=$V/libea_variants.so
echo "== attention v4: correctness"; timeout 180 python -m pytest tests/test_kernels_gpu.py -q -x -s -k "attention_v4" 2>&1 | grep -E "parity|passed|failed|rror|assert" | cut -c1-300 | tee gpurun_out/${TAG}_attention_v4_test.txt
echo "== four-wave GEMM, both schedules: correctness"; timeout 400 python -m pytest tests/test_kernels_gpu.py -q -x -k "gemm_w4" 2>&1 | tail -5 | tee gpurun_out/${TAG}_gemm_w4_test.txt
if grep -q passed gpurun_out/${TAG}_attention_v4_test.txt && ! grep -q failed gpurun_out/${TAG}_attention_v4_test.txt; then
  for lib in variants V4S2 V4F0 V4F2 V4H4 V4H12 V4P24 variants; do
    [ -f $V/libea_$lib.so ] || continue
    echo "== attention A/B, library $lib"; EA_LIB_PATH=$V/libea_$lib.so timeout 150 python tools/ab_attn_lib.py 2>&1 | grep -E "TFLOPs|v4_vs_v3" | cut -c1-260
  done | tee gpurun_out/${TAG}_attention_v4_ab.txt
fi
if grep -q passed gpurun_out/${TAG}_gemm_w4_test.txt && ! grep -q failed gpurun_out/${TAG}_gemm_w4_test.txt; then
  for lib in variants W4B24 W4B20E2; do
    [ -f $V/libea_$lib.so ] || continue
    echo "== GEMM A/B, library $lib"; EA_LIB_PATH=$V/libea_$lib.so timeout 240 python tools/ab_gemm_w4.py 2>&1 | grep -E "TFLOPs|bit_identical" | cut -c1-260
  done | tee gpurun_out/${TAG}_gemm_w4_ab.jsonl
fi
