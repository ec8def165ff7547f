#!/bin/bash
# SQ counters of the row-slab convolution kernels at the VAE's large-layer shapes: bash tools/run_pmc_conv.sh <outdir>
set -e
OUT=${1:-gpurun_out/pmc_conv}
REPO=$(pwd)
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU -d $REPO/$OUT/sq1 -o r -- python $REPO/tools/microbench_conv_rows.py > $REPO/$OUT/sq1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA -d $REPO/$OUT/sq2 -o r -- python $REPO/tools/microbench_conv_rows.py > $REPO/$OUT/sq2.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $REPO/$OUT/f -o r -- python $REPO/tools/microbench_conv_rows.py > $REPO/$OUT/f.log 2>&1
rocprofv3 --kernel-trace --stats -d $REPO/$OUT/kt -o r -- python $REPO/tools/microbench_conv_rows.py > $REPO/$OUT/kt.log 2>&1
cd $REPO
python - <<'PY'
import sqlite3, glob, sys, os
out = os.environ.get("OUTDIR", sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc_conv")
PY
for d in sq1 sq2 f; do
  db=$(find $OUT/$d -name "*.db" | head -1)
  python tools/rocprof_summary.py pmc_each $db conv3d > $OUT/$d.csv || true
done
python tools/rocprof_summary.py each $(find $OUT/kt -name "*.db" | head -1) conv3d > $OUT/kt.csv || true
rm -rf $OUT/sq1 $OUT/sq2 $OUT/f $OUT/kt
cat $OUT/kt.csv $OUT/sq1.csv $OUT/sq2.csv $OUT/f.csv
