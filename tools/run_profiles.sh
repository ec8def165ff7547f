#!/bin/bash
# rocprofv3 evidence for bench.py (config c3, 1 GPU): kernel trace + stats, then two separate PMC passes
# (FETCH_SIZE, WRITE_SIZE) as /opt/skills/guides/MI355X_MICROARCH.md prescribes.  Run on the GPU box:
#     bash tools/run_profiles.sh <tag>         -> gpurun_out/prof_<tag>/{kernel_stats.csv,pmc_*.csv,*.log}
# then copy the summaries into profiles/ (tools/rocprof_summary.py wrote them).
#     SKIP_PMC=1 bash tools/run_profiles.sh <tag>   -> the kernel trace only
# Every profiler run is under its own `timeout`: a process that dies under rocprofv3 (a GPU fault) can leave the tool waiting
# forever -- round 4 lost 25 GPU-minutes to exactly that.
set -e
TAG=${1:-r01b}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_under_rocprof.log 2>&1 || echo "kernel-trace run failed or timed out"
if [ -z "$SKIP_PMC" ]; then
  timeout 300 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o bench -- python $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-vae > $OUT/pmc_fetch.log 2>&1 || echo "FETCH_SIZE run failed or timed out"
  timeout 300 rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o bench -- python $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-vae > $OUT/pmc_write.log 2>&1 || echo "WRITE_SIZE run failed or timed out"
fi
cd $REPO
python tools/rocprof_summary.py stats $(find $OUT/trace -name "*.db" | head -1) > $OUT/kernel_stats.csv
if [ -z "$SKIP_PMC" ]; then
  python tools/rocprof_summary.py pmc $(find $OUT/pmc_fetch -name "*.db" | head -1) > $OUT/pmc_FETCH_SIZE.csv
  python tools/rocprof_summary.py pmc $(find $OUT/pmc_write -name "*.db" | head -1) > $OUT/pmc_WRITE_SIZE.csv
fi
[ -z "$SKIP_PMC" ] && python tools/update_traffic_json.py $OUT $TAG > $OUT/traffic_json.log 2>&1 || true
grep '"metric"' $OUT/bench_under_rocprof.log > $OUT/bench_under_rocprofv3.json || true
rm -rf $OUT/trace $OUT/pmc_fetch $OUT/pmc_write   # the sqlite databases are large; the summaries are what is kept
head -12 $OUT/kernel_stats.csv; [ -z "$SKIP_PMC" ] && { head -8 $OUT/pmc_FETCH_SIZE.csv; head -8 $OUT/pmc_WRITE_SIZE.csv; } || true
