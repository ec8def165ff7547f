#!/bin/bash
# schedule variants of the four-wave GEMM loop, one process per library, same box: FFN-up / FFN-down / QKV-width / fused QKV rows only
for rep in 1 2; do
for lib in base prog bar24 prog_bar24 prog_bar16; do
  if [ $lib = base ]; then unset EA_LIB_PATH; else export EA_LIB_PATH=$PWD/gpu_variants/libea_$lib.so; fi
  python tools/ab_gemm_w4a.py 1 2>/dev/null | grep "four-wave" | python -c "
import sys,json
rows=[json.loads(l) for l in sys.stdin]
print('$lib', ' '.join(f\"{r['what'].split()[0]}{r['what'].split()[1] if len(r['what'].split())>1 else ''}={r['TFLOPs']}\" for r in rows))"
done; done
