#!/bin/bash
# Round-5 counter evidence on the FINAL build, separate --pmc passes as MI355X_MICROARCH.md prescribes:
#   (1) the two GEMM kernels of this library next to the vendor kernel on the FFN-up shape (SQ waits / instruction mix / L2);
#   (2) the row-slab convolution kernels inside a REAL 49 x 1024^2 decode + encode (SQ_BUSY_CYCLES, MFMA ops, LDS waits, L2 hit / miss).
#     bash tools/run_pmc_r05.sh <outdir>
OUT=${1:-gpurun_out/pmc_r05}
REPO=$(pwd)
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {  # tag, script, counters...
  tag=$1; script=$2; shift 2
  timeout 240 rocprofv3 --pmc "$@" -d $REPO/$OUT/$tag -o r -- python $REPO/tools/$script > $REPO/$OUT/$tag.log 2>&1
  db=$(find $REPO/$OUT/$tag -name "*.db" | head -1)
  [ -n "$db" ] && python $REPO/tools/rocprof_summary.py pmc $db > $REPO/$OUT/$tag.csv
  rm -rf $REPO/$OUT/$tag
}
run gemm_sq1 prof_gemm_pair.py SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU
run gemm_sq2 prof_gemm_pair.py SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM SQ_WAVES
run vae_sq1 prof_vae_decode.py SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_LDS SQ_INSTS_MFMA
run vae_tc prof_vae_decode.py TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum
cd $REPO
for f in gemm_sq1 gemm_sq2 vae_sq1 vae_tc; do echo "== $f"; grep -h -v "^torch\|elementwise\|distribution" $OUT/$f.csv | head -40 | cut -c1-200; done
