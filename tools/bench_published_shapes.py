"""The reference's one published performance table, run shape by shape through the product pipeline.

/root/reference README.md:138-143: "The generation time for EasyAnimateV5.1-12B using different GPUs over 25 steps" -- whole
`EasyAnimatePipeline.__call__` wall time (and seconds per iteration) at six shapes on A10 / A100.  This tool times the same
thing here: the 12B transformer (48 MMDiT layers, d = 3072; random-init weights of the declared architecture), CFG on
(guidance 6, batch 2), 25 Flow-Euler steps, every step computed (TeaCache off), then the VAE decode of all frames and the
host copy of the numpy video `__call__` returns.  The text-encoder forward (Qwen2-VL-7B, once per call, `transformers`' own
code) is NOT in the timed region: prompt embeddings [1, 256, 3584] are passed in.

    python tools/bench_published_shapes.py [--shapes 384x672x49,768x1344x49] [--steps 25]

One JSON line per shape; `a100_*` are the README's numbers (other hardware, the reference's own code), `speedup_vs_a100` =
their total / this total.  A comparison across hardware, not a roofline statement -- bench.py carries that.
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch

# README.md:143 (A100 80GB row): shape -> (total seconds, seconds per iteration)
A100 = {"384x672x25": (45.0, 1.75), "384x672x49": (90.0, 3.7), "576x1008x25": (120.0, 4.7), "576x1008x49": (300.0, 11.4),
        "768x1344x25": (265.0, 10.6), "768x1344x49": (710.0, 28.3)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default=",".join(A100))
    ap.add_argument("--steps", type=int, default=25)
    a = ap.parse_args()
    import bench
    import bench_vae
    from easyanimate_amd import FlowMatchEulerDiscreteScheduler, _lib
    from easyanimate_amd.pipeline import EasyAnimatePipeline
    dev = torch.device("cuda:0")
    model = bench.build_model(48, dev)
    vae = bench_vae.build_vae()
    pipe = EasyAnimatePipeline(vae=vae, transformer=model, scheduler=FlowMatchEulerDiscreteScheduler(shift=1.0))
    g = torch.Generator(device="cpu").manual_seed(1)
    pos = torch.randn(1, 256, 3584, generator=g).to(dev, torch.bfloat16)
    neg = torch.randn(1, 256, 3584, generator=g).to(dev, torch.bfloat16)

    def call(h, w, f, steps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.no_grad():
            out = pipe(prompt_embeds=pos, negative_prompt_embeds=neg, video_length=f, height=h, width=w, num_inference_steps=steps,
                       guidance_scale=6.0, generator=torch.Generator(device="cpu").manual_seed(43), output_type="numpy")
        torch.cuda.synchronize()
        return time.perf_counter() - t0, out.frames

    for shape in a.shapes.split(","):
        h, w, f = (int(v) for v in shape.split("x"))
        call(h, w, f, 1)                               # first-use costs of this shape (allocator, packed weights)
        t2, _ = call(h, w, f, 2)
        torch.cuda.reset_peak_memory_stats()
        _lib.reset_counters()
        total, frames = call(h, w, f, a.steps)
        per_it = (total - t2) / (a.steps - 2)          # slope: the non-loop part (decode + host copy) cancels
        fl = (f - 1) // 4 + 1
        n_tok = fl * (h // 16) * (w // 16)
        ref_total, ref_it = A100.get(shape, (None, None))
        print(json.dumps({
            "shape": shape, "steps": a.steps, "video_tokens": n_tok, "total_s": round(total, 2), "s_per_it": round(per_it, 4),
            "decode_and_copy_s": round(total - per_it * a.steps, 3), "frames_shape": list(frames.shape),
            "finite": bool(torch.isfinite(torch.from_numpy(frames)).all()) if hasattr(frames, "shape") else None,
            "a100_total_s": ref_total, "a100_s_per_it": ref_it,
            "speedup_vs_a100": round(ref_total / total, 2) if ref_total else None,
            "peak_mem_GB": round(torch.cuda.max_memory_allocated() / 1e9, 1),
            "kernels": {k: v for k, v in sorted(_lib.counters().items()) if k.startswith(("attention", "conv_row", "conv_pp"))},
            "note": "12B random-init, CFG batch 2, TeaCache off, text-encoder forward not timed (embeddings passed in); A100 numbers: "
                    "reference README.md:143, the reference's own code on other hardware"}), flush=True)


if __name__ == "__main__":
    main()
