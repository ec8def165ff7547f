#!/bin/bash
# SQ counters of the attention kernel (own run, no trace domains): bash tools/run_pmc_attn.sh <variant> <outdir>
set -e
VAR=${1:-2}
OUT=${2:-gpurun_out/pmc_attn_v$VAR}
REPO=$(pwd)
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU -d $REPO/$OUT/sq1 -o r -- python $REPO/tools/prof_attn.py attn $VAR > $REPO/$OUT/sq1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA -d $REPO/$OUT/sq2 -o r -- python $REPO/tools/prof_attn.py attn $VAR > $REPO/$OUT/sq2.log 2>&1
cd $REPO
for d in sq1 sq2; do
  db=$(find $OUT/$d -name "*.db" | head -1)
  python tools/rocprof_summary.py pmc $db attention > $OUT/$d.csv || true
done
cat $OUT/sq1.csv $OUT/sq2.csv
