#!/bin/bash
# rocprofv3 kernel trace + stats of the VAE-only benchmark (config 4): bash tools/run_profiles_vae.sh <tag>
set -e
TAG=${1:-r01b}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_vae_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o vae -- python $REPO/tools/bench_vae.py --iters 1 > $OUT/vae_under_rocprof.log 2>&1
cd $REPO
python tools/rocprof_summary.py stats $(find $OUT/trace -name "*.db" | head -1) > $OUT/kernel_stats.csv
rm -rf $OUT/trace
head -14 $OUT/kernel_stats.csv
