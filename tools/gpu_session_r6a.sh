#!/bin/bash
# Round 6, first GPU session: the suite, the smoke canaries, LayerNorm sweep A/B, old-vs-new library A/B of the accumulator-output
# form of the four-wave kernels (GEMM shapes, VAE passes), one bench line.    bash tools/gpu_session_r6a.sh
mkdir -p gpurun_out/r6a
O=gpurun_out/r6a
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q -s > $O/suite.log 2>&1; echo "suite rc $?" >> $O/suite.log
tail -5 $O/suite.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?" >> $O/smoke.log
tail -6 $O/smoke.log
timeout 600 python tools/ab_layernorm.py 3 > $O/ab_layernorm.jsonl 2>&1
for v in cur old cur old; do
  if [ "$v" = cur ]; then unset EA_LIB_PATH; else export EA_LIB_PATH=$PWD/easyanimate_amd/lib/variants/libea_$v.so; fi
  timeout 600 python tools/ab_gemm.py 2>&1 | grep -v fp8 | sed "s/^/variant=$v /" >> $O/ab_gemm_accout.txt
  timeout 900 python tools/bench_vae.py --iters 2 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); print('variant=$v', json.dumps({k: {'MPix_per_s': r[k]['MPix_per_s'], 'seconds': r[k]['seconds']} for k in ('decode', 'encode')}))
" >> $O/ab_vae_accout.txt
done
unset EA_LIB_PATH
timeout 1200 python bench.py --steps 3 --warmup 1 --quick-cpu-baseline > $O/bench_c3.json 2> $O/bench_c3.err
tail -c 1500 $O/bench_c3.json
