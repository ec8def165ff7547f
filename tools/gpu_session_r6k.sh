#!/bin/bash
# Modelled 1 -> P curve on one GPU with this round's build: the per-rank compute of rank 0 in worlds of 2 / 4 / 8 (bench.py --emulate-rank),
# with the K / V^T exchange's bytes moved device-to-device under the own-slot pass, next to the plain N = 1 run of the same session.
mkdir -p gpurun_out/r6k
O=gpurun_out/r6k
export TMPDIR=/tmp
timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-vae > $O/n1.json 2> $O/n1.err
for P in 2 4 8; do
  timeout 900 python bench.py --steps 3 --warmup 1 --emulate-rank $P,0 --emulate-exchange > $O/emu_$P.json 2> $O/emu_$P.err
done
timeout 900 python bench.py --steps 3 --warmup 1 --emulate-rank 8,0 --emulate-exchange --sp-mode heads > $O/emu_8_heads.json 2> $O/emu_8_heads.err
python - <<PY
import json
n1 = json.load(open("$O/n1.json"))
print("N=1", n1["ms_per_step"])
for f in ("emu_2", "emu_4", "emu_8", "emu_8_heads"):
    try:
        r = json.load(open("$O/%s.json" % f))
        print(f, r["ms_per_step"], "speed-up bound", n1["ms_per_step"] / r["ms_per_step"], r["roofline"]["avg_launch_ms"], r["roofline"]["launches_per_block"])
    except Exception as ex:
        print(f, "failed", ex)
PY
