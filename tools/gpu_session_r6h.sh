#!/bin/bash
mkdir -p gpurun_out/r6h
O=gpurun_out/r6h
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_vae_gpu.py -m gpu -q -s -x -k "blocked" > $O/blocked_tests.log 2>&1; echo "rc $?" >> $O/blocked_tests.log
grep -E "dispatch|passed|failed|Error|rc " $O/blocked_tests.log | tail -8
timeout 900 python tools/ab_vae_blocked.py 3 2>&1 | grep -v amdgpu | tee $O/ab_vae_blocked.jsonl
