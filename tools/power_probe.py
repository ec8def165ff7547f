"""Socket power and shader clock while one kernel runs back to back (GPU box): python tools/power_probe.py
Polls `rocm-smi --showpower --showclocks --json` from a side thread during ~4 s loops of the config-3 attention launch
and of the big GEMMs.  Evidence for DESIGN.md 3.1 / 3.2: both kernels run power-limited, far below the 2.4 GHz peak clock."""
import json
import os
import subprocess
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from easyanimate_amd import ops


def poll(stop, out):
    while not stop.is_set():
        try:
            r = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=5)
            d = json.loads(r.stdout)
            card = next(iter(d.values()))
            rec = {}
            for k, v in card.items():
                kl = k.lower()
                if "power" in kl and "(w)" in kl:
                    rec["power_w"] = float(v)
                if kl.startswith("sclk clock speed"):
                    rec["sclk"] = v
            out.append(rec)
        except Exception as e:  # noqa
            out.append({"error": str(e)[:80]})
        time.sleep(0.15)


def run(name, fn, flop, seconds=4.0):
    fn(); torch.cuda.synchronize()
    stop, samples = threading.Event(), []
    th = threading.Thread(target=poll, args=(stop, samples))
    th.start()
    t0 = time.time(); n = 0
    while time.time() - t0 < seconds:
        for _ in range(5):
            fn()
        torch.cuda.synchronize(); n += 5
    dt = time.time() - t0
    stop.set(); th.join()
    pw = [s["power_w"] for s in samples if "power_w" in s]
    clk = [s["sclk"] for s in samples if "sclk" in s]
    print(json.dumps({"kernel": name, "TFLOPs": flop * n / dt / 1e12, "power_w_avg": sum(pw) / max(len(pw), 1), "power_w_max": max(pw or [0]),
                      "sclk_samples": clk[len(clk) // 2: len(clk) // 2 + 3], "n_samples": len(samples), "errors": [s for s in samples if "error" in s][:1]}), flush=True)


def main():
    dev = "cuda"
    B, H, S = 1, 48, 53504
    q = (torch.randn(B, H, S, 64, device=dev) * ops.FOLDED_Q_SCALE).to(torch.bfloat16)
    k = torch.randn(B, H, S, 64, device=dev).to(torch.bfloat16)
    vt = torch.randn(B, H, 64, S, device=dev).to(torch.bfloat16)
    out = torch.empty(B, S, H * 64, dtype=torch.bfloat16, device=dev)
    from easyanimate_amd import _lib
    zq = torch.zeros_like(q)
    for var in ((3, 2) if _lib.get_option("build_variants") == 1 else (3,)):
        _lib.set_option("attn_variant", var)
        run(f"attention v{var} c3 (B=1)", lambda: ops.attention(q, k, vt, S, ops.FOLDED_ATTN_SCALE, out=out), 4.0 * B * H * S * S * 64)
        run(f"attention v{var} c3, all-zero Q (same instruction stream, idle data)", lambda: ops.attention(zq, k, vt, S, ops.FOLDED_ATTN_SCALE, out=out), 4.0 * B * H * S * S * 64)
    _lib.set_option("attn_variant", 3)
    del q, k, vt, out, zq
    if "--attn-only" in sys.argv:
        return
    for (M, N, K) in [(8192, 8192, 8192), (53504, 12288, 3072)]:
        a = torch.randn(M, K, device=dev).to(torch.bfloat16)
        w = torch.randn(N, K, device=dev).to(torch.bfloat16)
        bias = torch.zeros(N, device=dev)
        c = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        run(f"gemm {M}x{N}x{K}", lambda: ops.gemm(a, w, bias, out=c), 2.0 * M * N * K)
        za = torch.zeros_like(a)
        run(f"gemm {M}x{N}x{K}, all-zero A", lambda: ops.gemm(za, w, bias, out=c), 2.0 * M * N * K)
        del a, w, c, za
    idle = []
    stop = threading.Event(); th = threading.Thread(target=poll, args=(stop, idle)); th.start(); time.sleep(1.0); stop.set(); th.join()
    print(json.dumps({"kernel": "idle", "samples": idle[-2:]}))


if __name__ == "__main__":
    main()
