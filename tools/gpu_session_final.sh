#!/bin/bash
# Round 6, final-build session: whole GPU suite, smoke, rocprofv3 kernel trace + the two PMC passes of bench.py (traffic stamp),
# SQ counters of the attention kernel, bench lines for configs 2 / 5 / 3.
mkdir -p gpurun_out/r6n
O=gpurun_out/r6n
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q -s > $O/suite.log 2>&1; echo "suite rc $?" >> $O/suite.log
tail -3 $O/suite.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?" >> $O/smoke.log
tail -2 $O/smoke.log
bash tools/run_profiles.sh r06n > $O/run_profiles.log 2>&1
cp gpurun_out/prof_r06n/*.csv gpurun_out/prof_r06n/*.json $O/ 2>/dev/null
timeout 900 python bench.py --config c2 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_c2.json 2> $O/bench_c2.err
timeout 900 python bench.py --config c5 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_c5.json 2> $O/bench_c5.err
timeout 1500 python bench.py --steps 5 --warmup 1 > $O/bench_c3.json 2> $O/bench_c3.err
python - <<PY
import json
for c in ("c2", "c5", "c3"):
    try:
        r = json.load(open("$O/bench_%s.json" % c))
        print(c, r["value"], r["ms_per_step"], r["roofline"]["avg_launch_ms"], r["roofline"].get("traffic"), (r.get("vae") or {}).get("decode_mpix_s"), (r.get("vae") or {}).get("encode_mpix_s"), (r.get("cpu_baseline") or {}).get("block_at_full_S_measured"))
    except Exception as ex:
        print(c, "failed", ex)
PY
