#!/bin/bash
# SQ / cache counters of this library's 256 x 256 GEMM next to the vendor kernel (hipBLASLt through torch.matmul) on the FFN-up shape:
#     bash tools/run_pmc_gemm.sh <outdir>        (separate --pmc passes, kernel-trace in its own run: MI355X_MICROARCH.md)
OUT=${1:-gpurun_out/pmc_gemm}
REPO=$(pwd)
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {  # tag, counters...
  tag=$1; shift
  timeout 120 rocprofv3 --pmc "$@" -d $REPO/$OUT/$tag -o r -- python $REPO/tools/prof_gemm_pair.py > $REPO/$OUT/$tag.log 2>&1
  db=$(find $REPO/$OUT/$tag -name "*.db" | head -1)
  [ -n "$db" ] && python $REPO/tools/rocprof_summary.py pmc $db > $REPO/$OUT/$tag.csv
  rm -rf $REPO/$OUT/$tag
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU
run sq2 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA
run sq3 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_MISC SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT
run tc TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum
timeout 120 rocprofv3 --kernel-trace --stats -d $REPO/$OUT/kt -o r -- python $REPO/tools/prof_gemm_pair.py > $REPO/$OUT/kt.log 2>&1
python $REPO/tools/rocprof_summary.py stats $(find $REPO/$OUT/kt -name "*.db" | head -1) > $REPO/$OUT/kt.csv
rm -rf $REPO/$OUT/kt
cd $REPO
grep -h -v "^torch\|elementwise\|distribution" $OUT/kt.csv $OUT/sq1.csv $OUT/sq2.csv $OUT/sq3.csv $OUT/tc.csv | cut -c1-220
