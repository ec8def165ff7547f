"""Writes profiles/attention_hbm_bytes_per_launch.json from the two PMC summaries of a `tools/run_profiles.sh <tag>` run, stamped
with the build it was measured on.  bench.py prints `roofline.traffic` only while that stamp equals easyanimate_amd/lib/build.sha256
(the field used to be a constant of an older build).

    python tools/update_traffic_json.py gpurun_out/prof_<tag> <tag>
"""
import datetime
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def per_launch(path, prefix):
    # kernel,counter,dispatches,sum,avg_per_dispatch,min,max -- the kernel name itself contains commas (template arguments)
    for line in open(path).read().splitlines()[1:]:
        if line.startswith(prefix):
            f = line.rsplit(",", 6)
            return float(f[4]), int(f[2])
    raise SystemExit(f"{path}: no kernel starting with {prefix}")


def main():
    d, tag = sys.argv[1], sys.argv[2]
    fetch, n_f = per_launch(os.path.join(d, "pmc_FETCH_SIZE.csv"), "attention_fwd_v3_kernel<0")
    write, n_w = per_launch(os.path.join(d, "pmc_WRITE_SIZE.csv"), "attention_fwd_v3_kernel<0")
    sha = open(os.path.join(ROOT, "easyanimate_amd", "lib", "build.sha256")).read().strip()
    B, H, S = 2, 48, 53504
    out = {
        "kernel": "attention_fwd_v3_kernel<0> (ea_attention_fwd_bf16, softmax scale folded into Q)",
        "workload": "bench.py config c3: B=2, H=48, S=53504, head_dim 64 (one launch per MMDiT block)",
        "source": f"profiles/{tag}_bench_c3_pmc_FETCH_SIZE.csv, profiles/{tag}_bench_c3_pmc_WRITE_SIZE.csv (rocprofv3 --pmc, separate passes, "
                  f"{n_f} / {n_w} dispatches)",
        "build_sha256": sha,
        "measured_utc": datetime.datetime.now(datetime.timezone.utc).strftime("%Y-%m-%d %H:%M"),
        "FETCH_SIZE_KB_per_launch": fetch,
        "WRITE_SIZE_KB_per_launch": write,
        "correction": "MI355X_MICROARCH.md HBM section: on gfx950 FETCH_SIZE reports 1/2 of the bytes of wide (16 B/lane) coalesced reads -> doubled; counters are in KiB",
        "hbm_bytes_per_launch": (2.0 * fetch + write) * 1024.0,
        "algorithmic_bytes_per_launch": 4 * B * H * S * 64 * 2,
    }
    # profiles/ is where bench.py reads it; the copy in the run directory is what travels back from the GPU box (gpurun_out/)
    for p in (os.path.join(ROOT, "profiles", "attention_hbm_bytes_per_launch.json"), os.path.join(d, "attention_hbm_bytes_per_launch.json")):
        json.dump(out, open(p, "w"), indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
