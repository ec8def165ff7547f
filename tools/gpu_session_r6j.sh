#!/bin/bash
# SQ counters of the three workgroup shapes of attention_fwd_v3_kernel (config-3 shape, three launches each), separate --pmc passes
export TMPDIR=/tmp
for sh in 4,2 8,2 4,3; do
  tag=$(echo $sh | tr , _)
  EA_PROF_ATTN_SHAPE=$sh timeout 600 bash tools/run_pmc_attn.sh 3 gpurun_out/r6j/pmc_attn_$tag > gpurun_out/r6j_$tag.log 2>&1
  rm -rf gpurun_out/r6j/pmc_attn_$tag/sq1 gpurun_out/r6j/pmc_attn_$tag/sq2
done
mkdir -p gpurun_out/r6j
mv gpurun_out/r6j_*.log gpurun_out/r6j/ 2>/dev/null
for t in 4_2 8_2 4_3; do echo "== $t"; cat gpurun_out/r6j/pmc_attn_$t/sq1.csv gpurun_out/r6j/pmc_attn_$t/sq2.csv | cut -d, -f1-6 | grep -v "^kernel"; done
