#!/bin/bash
mkdir -p gpurun_out/r6e
O=gpurun_out/r6e
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -s -x -k "eight_wave" > $O/attn_tests.log 2>&1; echo "rc $?" >> $O/attn_tests.log
tail -4 $O/attn_tests.log
EA_AB_NW=1 timeout 600 python tools/ab_attn_lib.py > $O/ab_attn_shapes.jsonl 2>&1
cat $O/ab_attn_shapes.jsonl | grep -v amdgpu
for st in 2 3 2 3; do
  timeout 900 python - <<PY >> $O/bench_stages.txt 2>&1
import sys
sys.path.insert(0, '.')
from easyanimate_amd import _lib
_lib.set_option("attn_stages", $st)
sys.argv = ["bench.py", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-vae"]
import runpy
runpy.run_path("bench.py", run_name="__main__")
PY
done
grep -o '"value": [0-9.]*\|"avg_launch_ms": [0-9.]*' $O/bench_stages.txt
