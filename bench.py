#!/usr/bin/env python
"""Headline benchmark: denoise-steps/s of the EasyAnimate V5.1 12B MMDiT at 49 frames x 1024^2 (BASELINE.json
`metric`) on N MI355X GPUs of one node.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

One *step* = one iteration of the sampling loop (pipeline_easyanimate.py:1069-1111): transformer forward on the
CFG pair (B=2) + CFG combine + Flow-matching Euler update, TeaCache off.  Synthetic data: random-init weights of
the declared 12B architecture, N(0,1) latents and text embeddings.  N>1 = sequence parallel
over the video tokens (strong scaling: the job is one video).  Rank 0 prints ONE JSON line.  At N=1 the same line carries
the second half of the metric, "vae": decode / encode MPix/s of the causal 3-D VAE at 49 x 1024^2 (config 4), timed in
the same process after the DiT steps.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0  # dense MFMA bf16 peak, /opt/skills/guides/MI355X_MICROARCH.md

CONFIGS = {
    # name: (layers, frames, height, width)   -- all: d=3072 (48x64), ff 12288, text 256 x 3584, CFG batch 2
    "c3": dict(desc="12B DiT (48 MMDiT layers, d=3072), 49f x 1024x1024, T=256 text tokens, CFG batch 2, bf16",
               layers=48, frames=49, height=1024, width=1024),
    "c2": dict(desc="7B-class DiT (28 MMDiT layers, d=3072), 49f x 512x512, T=256, CFG batch 2, bf16",
               layers=28, frames=49, height=512, width=512),
    "c5": dict(desc="12B InP DiT (48 MMDiT layers, d=3072, in_channels 33), 49f x 768x768, T=256, CFG batch 2, bf16, "
                    "random inpaint latents in the loop; the one VAE encode of the conditioning video is timed separately "
                    "(i2v_conditioning) -- BASELINE config 5 on ONE of its four GPUs",
               layers=48, frames=49, height=768, width=768, in_channels=33),
    "tiny": dict(desc="2-layer d=3072 DiT, 5f x 128x128 (debug)", layers=2, frames=5, height=128, width=128),
    "small": dict(desc="2-layer d=3072 DiT, 9f x 256x256 = 768 video tokens (debug: enough tokens for 8 sequence shards)",
                  layers=2, frames=9, height=256, width=256),
}


def build_model(layers: int, device, in_channels: int = 16):
    from easyanimate_amd import EasyAnimateTransformer3DModel
    with torch.device("meta"):
        m = EasyAnimateTransformer3DModel(num_attention_heads=48, attention_head_dim=64, in_channels=in_channels, out_channels=16,
                                          patch_size=2, num_layers=layers, time_embed_dim=512, add_norm_text_encoder=True,
                                          text_embed_dim=3584, text_embed_dim_t5=None, norm_eps=1e-5,
                                          time_position_encoding_type="3d_rope", enable_text_attention_mask=True)
    m = m.to(torch.bfloat16).to_empty(device=device)
    torch.cuda.manual_seed(0)  # identical weights on every rank
    with torch.no_grad():
        for name, p in m.named_parameters():
            if p.dim() >= 2:
                fan_in = p[0].numel()
                p.uniform_(-1.0 / math.sqrt(fan_in), 1.0 / math.sqrt(fan_in))
            elif name.endswith("bias"):
                p.uniform_(-0.02, 0.02)
            else:  # norm gains
                p.fill_(1.0)
    return m.eval()


def _block_shapes(d=3072):
    shapes = {}
    for n in ("norm1", "norm2"):
        shapes.update({f"{n}.linear.weight": (6 * d, 512), f"{n}.linear.bias": (6 * d,), f"{n}.norm.weight": (d,), f"{n}.norm.bias": (d,)})
    for a in ("attn1", "attn2"):
        for l in ("to_q", "to_k", "to_v", "to_out.0"):
            shapes.update({f"{a}.{l}.weight": (d, d), f"{a}.{l}.bias": (d,)})
        for l in ("norm_q", "norm_k"):
            shapes.update({f"{a}.{l}.weight": (64,), f"{a}.{l}.bias": (64,)})
    for f in ("ff", "txt_ff"):
        shapes.update({f"{f}.net.0.proj.weight": (4 * d, d), f"{f}.net.0.proj.bias": (4 * d,), f"{f}.net.2.weight": (d, 4 * d), f"{f}.net.2.bias": (d,)})
    return shapes


def cpu_baseline(S_bench: int, grid, quick: bool = False, budget_s: float = 25.0, full_limit_s: float = 400.0):
    """One MMDiT block of the benchmark shape on the host cores, fp32, B = 1 -- the UNCHANGED reference block
    (easyanimate/models/attention.py:1028-1163 through oracle/ref_loader, kind "reference") where /root/reference exists (the build
    container), else the oracle restatement (oracle/restatement.dit_block, kind "port"; measured bit-identical to the reference
    block and equally fast, profiles/r04b_cpu_reference_vs_port.json).  SURVEY 8(d): time ONE BLOCK AT FULL S and report L x that.

    Two stages: (1) a bounded sample -- one block at 1024 video + 256 text tokens and the SDPA alone at S = 4096, extrapolated
    separately (linear in S / quadratic in S) -- which also warms the thread pool and predicts the full-S time; (2) unless `quick`
    or the prediction exceeds full_limit_s, ONE block at the FULL sequence length (grid = (frames, h, w) patches; S = 53 504 at
    config 3), measured, not extrapolated.  Returns (seconds per block and sample, kind, measured_at_full_S, threads, description)."""
    import torch.nn.functional as F
    from easyanimate_amd.synthetic import synth_state_dict
    from oracle import ref_loader
    from oracle import restatement as R
    threads = os.cpu_count() or 1
    torch.set_num_threads(threads)
    d, H, T = 3072, 48, 256
    sd = synth_state_dict(_block_shapes(d), 0)
    crops = ((0, 8), (30, 38))
    kind, blk, shim = "port", None, None
    if ref_loader.available():
        try:
            ns = ref_loader.load()
            blk = ns.attention.EasyAnimateDiTBlock(dim=d, num_attention_heads=H, attention_head_dim=64, time_embed_dim=512, norm_eps=1e-5,
                                                   is_mmdit_block=True).eval()
            blk.load_state_dict(sd, strict=True)
            kind, shim = "reference", ns.shim
        except Exception as ex:     # noqa: BLE001
            print(f"[cpu_baseline] reference block not usable ({ex!r}); timing the port", file=sys.stderr)
            blk = None

    def block_fn(frames, gh, gw, seed):
        g = torch.Generator().manual_seed(seed)
        n = frames * gh * gw
        h, e, temb = torch.randn(1, n, d, generator=g), torch.randn(1, T, d, generator=g), torch.randn(1, 512, generator=g)
        if blk is not None:
            rope = shim.get_3d_rotary_pos_embed(64, crops, (gh, gw), frames, use_real=True)
            return lambda: blk(h, e, temb, image_rotary_emb=rope)
        rope = R.rope_3d(64, crops, (gh, gw), frames)
        return lambda: R.dit_block(sd, "", h, e, temb, rope, H, 1e-5)

    def timed(fn, budget, max_reps=3):
        fn()  # warm-up
        t0 = time.perf_counter()
        reps = 0
        while reps < max_reps and time.perf_counter() - t0 < budget:
            fn()
            reps += 1
        return (time.perf_counter() - t0) / max(reps, 1), reps

    N = 1024
    S_s, S_a = T + N, 4096
    g = torch.Generator().manual_seed(0)
    qs = torch.randn(1, H, S_s, 64, generator=g)
    qa = torch.randn(1, H, S_a, 64, generator=g)
    with torch.no_grad():
        t_blk, r1 = timed(block_fn(1, 32, 32, 0), budget_s * 0.5)
        t_att_s, _ = timed(lambda: F.scaled_dot_product_attention(qs, qs, qs), budget_s * 0.1)
        t_att_a, r2 = timed(lambda: F.scaled_dot_product_attention(qa, qa, qa), budget_s * 0.4)
    per_token = max(t_blk - t_att_s, 0.0) / S_s
    per_s2 = t_att_a / (S_a * S_a)
    t_est = per_token * S_bench + per_s2 * S_bench * S_bench
    what = ("the unchanged reference EasyAnimateDiTBlock (attention.py:1028-1163 via oracle/ref_loader)" if kind == "reference"
            else "oracle/restatement.dit_block (the port; /root/reference is absent on this box)")
    sample = (f"{what}, fp32, B=1, {threads} threads. Bounded sample: one block at {N} video + {T} text tokens ({r1} reps, {t_blk:.2f} s; its SDPA "
              f"{t_att_s:.2f} s) -> {per_token * 1e3:.3f} ms per token; SDPA alone at S={S_a} ({r2} reps, {t_att_a:.2f} s) -> {per_s2 * 1e9:.3f} ns "
              f"per S^2; extrapolated to S={S_bench}: {t_est:.1f} s per block and sample")
    if quick or t_est > full_limit_s:
        why = "--quick-cpu-baseline" if quick else f"the predicted full-S block time exceeds {full_limit_s:.0f} s on these {threads} cores"
        return t_est, kind, False, threads, sample + f" -- REPORTED VALUE IS THIS EXTRAPOLATION ({why})"
    with torch.no_grad():
        fn = block_fn(*grid, 1)
        t0 = time.perf_counter()
        fn()
        t_full = time.perf_counter() - t0
    return t_full, kind, True, threads, (f"ONE BLOCK AT THE FULL SEQUENCE LENGTH S={S_bench} MEASURED: {t_full:.1f} s (one pass, after the warm "
                                         f"bounded sample; the sample's extrapolation predicted {t_est:.1f} s). " + sample)


def vae_cpu_baseline(quick: bool = False):
    """The VAE on the host cores, fp32, full width: the unchanged reference AutoencoderKLMagvit (kind "reference", build container) or
    the oracle restatement (oracle/restatement_vae.py, kind "port") at 9 x 256^2 (quick: 5 x 192^2), after one warm pass
    at 5 x 64^2.  MPix/s of decode and encode; the rate is what is reported."""
    from easyanimate_amd import AutoencoderKLMagvit
    from easyanimate_amd.synthetic import synth_state_dict
    from oracle import ref_loader
    from oracle import restatement_vae as RV
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_vae
    # oneDNN's direct 3-D convolutions at these sizes get SLOWER beyond a few dozen threads (measured on the 256-core host of round 4:
    # 32 threads beat 256), hence the cap; the DiT leg (GEMM / SDPA bound) uses every core
    threads = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(threads)
    with torch.device("meta"):
        shapes = {k: tuple(v.shape) for k, v in AutoencoderKLMagvit(**bench_vae.FULL).state_dict().items()}
    sd = synth_state_dict(shapes, 2)
    kind, vae = "port", None
    if ref_loader.available():
        try:
            ns = ref_loader.load()
            vae = ns.autoencoder_magvit.AutoencoderKLMagvit(**bench_vae.FULL).eval()
            vae.load_state_dict(sd, strict=True)
            kind = "reference"
        except Exception as ex:     # noqa: BLE001
            print(f"[vae_cpu_baseline] reference VAE not usable ({ex!r}); timing the port", file=sys.stderr)
            vae = None
    dec = (lambda z: vae.decode(z)[0]) if vae is not None else (lambda z: RV.vae_decode(sd, z, 32))
    enc = (lambda v: vae.encode(v)[0].mode()) if vae is not None else (lambda v: RV.vae_encode_moments(sd, v, 32))
    frames, size = (5, 192) if quick else (9, 256)
    g = torch.Generator().manual_seed(9)
    mk = lambda f, s_: (torch.rand(1, 3, f, s_, s_, generator=g) * 2 - 1, torch.randn(1, 16, (f - 1) // 4 + 1, s_ // 8, s_ // 8, generator=g))
    with torch.no_grad():
        wv, wz = mk(5, 64)
        dec(wz), enc(wv)            # warm pass
        video, z = mk(frames, size)
        t0 = time.perf_counter()
        dec(z)
        t_dec = time.perf_counter() - t0
        t0 = time.perf_counter()
        enc(video)
        t_enc = time.perf_counter() - t0
    mpix = frames * size * size / 1e6
    what = "the unchanged reference AutoencoderKLMagvit via oracle/ref_loader" if kind == "reference" else "oracle/restatement_vae (port)"
    return {"decode_mpix_s": mpix / t_dec, "encode_mpix_s": mpix / t_enc, "unit": "MPix/s", "cores": threads, "kind": kind,
            "sample": f"{what}, fp32, full width, {frames} x {size}^2 (decode {t_dec:.1f} s, encode {t_enc:.1f} s; one timed pass each after a "
                      f"warm pass at 5 x 64^2); {threads} threads: oneDNN's 3-D convolutions slow down beyond a few dozen threads, so this leg "
                      f"is capped at 32 while the DiT leg uses every core"}


def vae_section(cpu: bool, quick: bool = False):
    """BASELINE.json metric, second half: VAE decode / encode MPix/s at 49 x 1024^2 (config 4), same process, after the
    DiT steps.  Inputs are resident in HBM when the timed region starts."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_vae
    vae = bench_vae.build_vae()
    r = bench_vae.run(vae, 49, 1024, iters=2, kernel_breakdown=True)
    dec, enc = r["decode"], r["encode"]
    out = {"workload": "AutoencoderKLMagvit (V5/V5.1 widths 128/256/512/512), 49 x 1024 x 1024 RGB <-> latents [16,13,128,128], bf16, "
                       "random-init weights; decode randn/scaling_factor, encode U(-1,1) -> .mode()",
           "decode_mpix_s": dec["MPix_per_s"], "encode_mpix_s": enc["MPix_per_s"], "unit": "MPix/s",
           "decode_s": dec["seconds"], "encode_s": enc["seconds"],
           "bound": "mfma (3x3x3 convolutions, AI ~1300 FLOP/B); hbm_frac is the minimal activation traffic over 8 TB/s",
           "mfma_frac": dec["mfma_frac"], "hbm_frac": dec["hbm_frac"],
           "executed_mfma_frac": dec["executed_mfma_frac"], "flop_decode_executed": dec["executed_flop"],
           "note": "mfma_frac = ALGORITHMIC FLOPs of the reference's decode over time; the up-samplers run in sub-pixel form (12 of 27 taps) and "
                   "the layers behind a virtual temporal x2 with merged temporal taps (18 of 27), so fewer MFMA FLOPs are issued: "
                   "executed_mfma_frac counts those",
           "encode_mfma_frac": enc["mfma_frac"], "encode_hbm_frac": enc["hbm_frac"],
           "flop_decode": dec["algorithmic_flop"], "flop_encode": enc["algorithmic_flop"],
           "dominant_kernel": dec["dominant_kernel"], "avg_launch_ms": dec["dominant_avg_launch_ms"],
           "dominant_share_of_decode": dec["dominant_share_of_pass"],
           "decode_conv_kernels_ms": dec["conv_kernels_ms"], "encode_conv_kernels_ms": enc["conv_kernels_ms"],
           "finite_output": dec["finite"] and enc["finite"], "peak_mem_GB": r["peak_mem_GB"]}
    del vae
    torch.cuda.empty_cache()
    if cpu:
        out["cpu_baseline"] = vae_cpu_baseline(quick)
    return out


def self_launch(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: re-run the same command line as N ranks of one node under
    torch.distributed.run (--master-addr 127.0.0.1, a free port).  Returns the launcher's exit code."""
    import socket
    import subprocess
    shared = os.environ.get("EA_BENCH_SHARED_DEVICE") == "1"
    have = torch.cuda.device_count()
    if have < n and not shared:
        print(f"bench.py --gpus {n}: this node exposes {have} GPU(s); one rank per GPU is required "
              f"(EA_BENCH_SHARED_DEVICE=1 runs the ranks on one device over gloo, for testing the code path only)", file=sys.stderr)
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: what RCCL needs on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 1) // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    return subprocess.run(cmd, env=env).returncode


_STAGE = {"stage": "start"}


def _stage(name: str) -> None:
    _STAGE["stage"] = name


def _error_line(args, world: int, rank: int, err: str) -> str:
    """What rank 0 prints INSTEAD of the result line when a run dies (VERDICT r5 next #3b): the same keys a reader of the result
    line looks for, value null, the stage the rank was in and the error -- one JSON line, not a launcher traceback."""
    return json.dumps({"metric": "denoise-steps/sec (49f x 1024^2, 12B DiT)" if args.config == "c3" else f"denoise-steps/sec ({args.config})",
                       "value": None, "unit": "denoise-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                       "error": err, "stage": _STAGE["stage"], "rank": rank,
                       "env": {k: os.environ.get(k) for k in ("EA_SP_MODE", "EA_SP_GROUPS", "EA_SP_INPLACE", "NCCL_MAX_NCHANNELS")}})


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="c3", choices=sorted(CONFIGS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--quick-cpu-baseline", action="store_true",
                    help="cpu_baseline from the bounded sample only (extrapolated to the benchmark's S) instead of one block measured at full S")
    ap.add_argument("--emulate-rank", default=None, metavar="P,r",
                    help="one GPU runs the per-step COMPUTE of rank r in a world of P ranks (no communication): a labelled "
                         "MODEL of the scaling curve, never a measurement of it")
    ap.add_argument("--emulate-exchange", action="store_true",
                    help="with --emulate-rank: also move the bytes the per-block K / V^T all-gather would bring in (P'-1 slots) "
                         "device-to-device on a side stream under the own-slot attention pass (HBM / copy-engine contention "
                         "enters the model; xGMI link time does not)")
    ap.add_argument("--sp-mode", default=os.environ.get("EA_SP_MODE", "keys"), choices=["keys", "heads"],
                    help="multi-GPU exchange of the full-attention blocks: 'keys' = in-place K / V^T all-gather under the own-slot "
                         "attention pass (default), 'heads' = head all-to-all around one contiguous attention launch")
    ap.add_argument("--no-vae", action="store_true", help="skip the VAE half of the metric (config 4), timed after the DiT steps at N=1")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N`: start the N ranks ourselves (one process per GPU, RCCL rendezvous on the
        # loopback address) by re-executing this file under torch.distributed.run; rank 0's JSON line is the output.
        raise SystemExit(self_launch(args.gpus))
    if world != args.gpus:
        args.gpus = world
    # EA_BENCH_SHARED_DEVICE=1 (testing only, 1-GPU box): every rank uses cuda:0 and the rendezvous is gloo -- RCCL
    # refuses two ranks on one device; the timing of such a run means nothing, it exercises the multi-rank code path.
    if world > 1:
        # a multi-rank run that dies must still leave ONE JSON line on rank 0's stdout: exceptions on this rank are caught below;
        # when ANOTHER rank dies first the launcher sends SIGTERM to the survivors, which rank 0 turns into the same line
        import signal
        import traceback

        def on_term(signum, frame):
            if rank == 0:
                print(_error_line(args, world, rank, "terminated by the launcher (SIGTERM): another rank failed first -- its "
                                                     "traceback is on stderr"), flush=True)
            os._exit(1)
        signal.signal(signal.SIGTERM, on_term)
        try:
            return _run(args, world, rank, local_rank)
        except BaseException as ex:      # noqa: BLE001  (SystemExit included: argument errors inside _run)
            if isinstance(ex, SystemExit) and ex.code in (0, None):
                raise
            traceback.print_exc()
            line = _error_line(args, world, rank, f"{type(ex).__name__}: {ex}")
            print(line, file=sys.stdout if rank == 0 else sys.stderr, flush=True)
            if rank != 0:
                try:    # hand the line to rank 0's watchdog (it may sit in a collective's C++ wait, where no signal handler runs)
                    import torch.distributed as dist
                    dist.distributed_c10d._get_default_store().set("ea_bench_error", line)
                    time.sleep(2.0)
                except Exception:   # noqa: BLE001
                    pass
            os._exit(1)                 # no destroy_process_group: the other ranks may be inside a collective
    return _run(args, world, rank, local_rank)


def _run(args, world: int, rank: int, local_rank: int):
    shared = os.environ.get("EA_BENCH_SHARED_DEVICE") == "1"
    if shared:
        local_rank = 0
    _stage("set_device")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL's all-gather runs beside an attention kernel that keeps two workgroups on every CU: one channel = one
        # workgroup, so NCCL_MAX_NCHANNELS bounds the CUs taken away from it.  It is deliberately NOT set here: too few channels
        # stretch the exchange beyond the own-slot pass that hides it (every exposed ms is paid 48 times per step), too many
        # cost the attention kernel a few percent for ~2 ms per block -- the asymmetric risk favours RCCL's own default until
        # a multi-GPU node has measured the knee
        import datetime
        _stage("init_process_group")
        # a hang must end in minutes, not in the driver's timeout: 5 minutes cover RCCL's lazy communicator set-up on 8 ranks
        tmo = datetime.timedelta(seconds=int(os.environ.get("EA_BENCH_COLLECTIVE_TIMEOUT_S", "300")))
        if shared:
            dist.init_process_group("gloo", timeout=tmo)
        else:
            dist.init_process_group("nccl", device_id=device, timeout=tmo)
        if rank == 0:
            # watchdog: another rank that dies hands its error line over through the rendezvous store; this thread prints it as
            # rank 0's ONE JSON line even while the main thread is blocked inside a collective, then ends the process
            import threading
            store = dist.distributed_c10d._get_default_store()

            def watch():
                while True:
                    time.sleep(0.5)
                    try:
                        if store.check(["ea_bench_error"]):
                            print(store.get("ea_bench_error").decode(), flush=True)
                            os._exit(1)
                    except Exception:   # noqa: BLE001  (store gone: the run is over)
                        return
            threading.Thread(target=watch, daemon=True).start()

    from easyanimate_amd import FlowMatchEulerDiscreteScheduler, ops, sequence_parallel
    from easyanimate_amd.pipeline import EasyAnimatePipeline

    cfg = CONFIGS[args.config]
    _stage("build_model")
    model = build_model(cfg["layers"], device, cfg.get("in_channels", 16))
    if world > 1:
        _stage("sequence_parallel.enable (sub-groups)")
        sequence_parallel.enable(model, mode=args.sp_mode)
    emu = None
    if args.emulate_rank:
        if world != 1:
            raise SystemExit("--emulate-rank is a single-process mode")
        emu = tuple(int(v) for v in args.emulate_rank.split(","))
        model.sequence_parallel = sequence_parallel.EmulatedRank(*emu)
        model.sequence_parallel.emulate_exchange = bool(args.emulate_exchange)
        model.sequence_parallel.mode = args.sp_mode
        args.no_vae = args.no_cpu_baseline = True
    sched = FlowMatchEulerDiscreteScheduler(shift=1.0)
    pipe = EasyAnimatePipeline(vae=None, transformer=model, scheduler=sched)

    K, W = args.steps, args.warmup
    n_sched = max(50, K + W)
    sched.set_timesteps(n_sched, device=device, mu=1)
    g = torch.Generator(device="cpu").manual_seed(43)
    latents = pipe.prepare_latents(1, 16, cfg["frames"], cfg["height"], cfg["width"], torch.bfloat16, device, g)
    embeds = torch.randn(2, 256, 3584, generator=torch.Generator(device="cpu").manual_seed(1)).to(device, torch.bfloat16)
    rope = pipe.rotary_embedding(cfg["height"], cfg["width"], latents.size(2))
    inpaint = None
    if cfg.get("in_channels", 16) > 16:
        inpaint = torch.randn((2, cfg["in_channels"] - 16) + tuple(latents.shape[2:]), generator=torch.Generator(device="cpu").manual_seed(2)).to(device, torch.bfloat16)
    Fl, hl, wl = latents.shape[2:]
    N_tok = Fl * (hl // 2) * (wl // 2)
    S = 256 + N_tok
    B, H, d, L = 2, 48, 3072, cfg["layers"]

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if os.environ.get("EA_BENCH_FAULT_RANK") == str(rank):     # tests: this rank dies in its first forward
        def _boom(*a, **k):
            raise RuntimeError("EA_BENCH_FAULT_RANK: injected failure in the first forward")
        model.forward = _boom
    with torch.no_grad():
        _stage("warmup (first forward: RCCL communicators, first-use check of the in-place all-gather)" if world > 1 else "warmup")
        if W > 0:
            latents = pipe.denoise(latents, embeds, rope, sched.timesteps[:W], 6.0, inpaint_latents=inpaint)
        sync()
        _stage("timed steps")
        if world > 1:
            model.sequence_parallel.profile_wait = True    # HIP events around every stream-level wait for a collective
        t0 = time.perf_counter()
        with ops.KernelTimer("attention") as kt:
            latents = pipe.denoise(latents, embeds, rope, sched.timesteps[W:W + K], 6.0, inpaint_latents=inpaint)
        sync()
        elapsed = time.perf_counter() - t0
    elapsed_own = elapsed
    _stage("report")
    if world > 1:
        tmax = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = tmax.item()
    finite = bool(torch.isfinite(latents.float()).all().item())
    gathered_devices = [None] * world
    rank_reports = [None] * world
    if world > 1:   # which device each rank really ran on (a shared-device test run shows up here)
        dist.all_gather_object(gathered_devices, f"cuda:{local_rank}" + ("(shared)" if shared else ""))
        # self-check of the first hardware run: every rank steps the SAME gathered prediction through the same scheduler
        # kernel, so the final latents must be bit-identical on all ranks; and where the compute stream waited for a collective
        import hashlib
        waits = model.sequence_parallel.exposed_wait_ms()
        dist.all_gather_object(rank_reports, {
            "latents_sha1": hashlib.sha1(latents.detach().float().cpu().numpy().tobytes()).hexdigest(),
            "exposed_wait_ms_per_step": {k: round(t / K, 3) for k, (n, t) in waits.items()},
            "waits_per_step": {k: n / K for k, (n, t) in waits.items()}, "step_ms": round(elapsed_own / K * 1e3, 2)})

    # ---- roofline of the dominant kernel (attention forward): algorithmic FLOPs per block / HIP-event duration.
    # Multi-GPU: a rank runs its batch slice (CFG axis) and its query shard (sequence axis) as two or three key-range
    # launches per block (local keys while the K/V all-gather is in flight, then the remote keys): they are summed.
    durs = kt.durations_ms()
    n_blocks = L * K
    if world > 1 or emu:
        sp = model.sequence_parallel
        b_loc = 1 if sp.axis.cfg_degree == 2 else B
        lo, hi = sp.shard_range()
        q_rows = 256 + (hi - lo)
        par = (f"cfg{sp.axis.cfg_degree} x sp{sp.size}: CFG pair split over rank halves, video-token sequence parallel "
               f"inside a half; " + ("K | V projected first, in-place asynchronous K/V^T all-gather under the Q projection and the "
               f"own-slot attention pass" if sp.mode == "keys" else "head all-to-all (q, k, v^T out; o back) around one contiguous attention "
               f"launch over H / P' heads"))
    else:
        b_loc, q_rows, par = B, S, "single GPU"
    att_ms = sum(durs) / max(n_blocks, 1)
    att_flop = 4.0 * q_rows * S * 64 * b_loc * H
    achieved = att_flop / (att_ms * 1e-3) / 1e12 if att_ms > 0 else 0.0
    flop_step = B * L * (24.0 * S * d * d + 4.0 * S * S * d)
    traffic = None
    pmc = os.path.join(ROOT, "profiles", "attention_hbm_bytes_per_launch.json")
    if os.path.exists(pmc) and world == 1 and args.config == "c3":
        try:   # HBM bytes per attention launch from the PMC passes -- only if they were taken on THIS build (tools/update_traffic_json.py)
            rec = json.load(open(pmc))
            sha = open(os.path.join(ROOT, "easyanimate_amd", "lib", "build.sha256")).read().strip()
            traffic = rec.get("hbm_bytes_per_launch") if rec.get("build_sha256") == sha else None
        except Exception:
            traffic = None

    out = {
        "metric": "denoise-steps/sec (49f x 1024^2, 12B DiT)" if args.config == "c3" else f"denoise-steps/sec ({args.config})",
        "value": K / elapsed,
        "unit": "denoise-steps/s",
        "n_gpus": world,
        "steps": K,
        "warmup": W,
        "ms_per_step": elapsed / K * 1e3,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "bf16",
        "data": "synthetic (random-init weights of the declared 12B architecture, N(0,1) latents + text embeddings)",
        "config": {"workload": cfg["desc"], "video_tokens": N_tok, "seq_len": S, "cfg_batch": 2,
                   "parallelism": par,
                   "flop_per_step": flop_step, "step_mfma_frac": flop_step * K / elapsed / (PEAK_BF16_TFLOPS * 1e12 * world),
                   "finite_output": finite},
        "roofline": {"bound": "mfma", "kernel": "attention_fwd_v3_kernel (ea_attention_fwd_bf16 / _range_bf16)",
                     "achieved": achieved, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_BF16_TFLOPS,
                     "traffic": traffic,
                     "traffic_source": None if traffic is None else "profiles/attention_hbm_bytes_per_launch.json: rocprofv3 --pmc FETCH_SIZE / "
                                       "WRITE_SIZE passes of a builder run on this build (sha-matched), replayed -- not measured in this run",
                     "flop_per_launch": att_flop, "avg_launch_ms": att_ms,
                     "launches_timed": len(durs), "launches_per_block": len(durs) / max(n_blocks, 1)},
    }
    if world > 1:
        out["rccl"] = {"ranks_seen": dist.get_world_size(), "backend": dist.get_backend(),
                       "devices": sorted(set(gathered_devices)), "launched_by": os.environ.get("TORCHELASTIC_RUN_ID", "external")}
        out["rank_agreement"] = len({r["latents_sha1"] for r in rank_reports}) == 1
        try:
            rccl_version = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:   # noqa: BLE001
            rccl_version = None
        out["rccl"].update({"rccl_version": rccl_version, "NCCL_MAX_NCHANNELS": os.environ.get("NCCL_MAX_NCHANNELS"),
                            "NCCL_MIN_NCHANNELS": os.environ.get("NCCL_MIN_NCHANNELS"),
                            "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")})
        out["exchange"] = {"mode": model.sequence_parallel.mode,
                           # false = the first-use check of the in-place all-gather failed on some rank and every rank fell back
                           "inplace": bool(model.sequence_parallel.inplace), "inplace_requested": bool(model.sequence_parallel.inplace_requested),
                           "head_groups": model.sequence_parallel.head_groups(H) if model.sequence_parallel.mode == "keys" else 1,
                           "what": "per rank: time the compute stream spent between two HIP events around each wait for a collective "
                                   "(the EXPOSED part of the exchange), per denoise step; step_ms = the rank's own wall time per step",
                           "per_rank": [{k: v for k, v in r.items() if k != "latents_sha1"} for r in rank_reports]}
    if emu:
        P, r = emu
        out["metric"] = f"MODELLED per-rank compute-side rate, rank {r} of {P} (not a measurement of {P} GPUs)"
        out["emulated"] = {"world": P, "rank": r, "what": "this rank's batch slice (CFG axis), token shard through every per-token "
                           "kernel, and its queries over all keys in the two-pass local / remote form with random remote K / V^T; "
                           "collectives and their overlap with the own-slot pass are NOT included"
                                   + (" -- except that the bytes of the K / V^T all-gather are moved device-to-device on a side stream "
                                      "under the own-slot pass (--emulate-exchange)" if args.emulate_exchange else ""),
                           "exchange_emulated": bool(args.emulate_exchange),
                           "head_groups": model.sequence_parallel.head_groups(H) if model.sequence_parallel.mode == "keys" else 1,
                           "speedup_upper_bound_needs": "T_1 / this ms_per_step, T_1 from the plain N=1 run of the same session"}
        out["config"]["step_mfma_frac"] = flop_step * K / elapsed / (PEAK_BF16_TFLOPS * 1e12 * P)
    if rank == 0 and world == 1 and not args.no_vae and args.config == "c3":
        # the other half of BASELINE.json's metric: the DiT is released first (the VAE's activations peak at ~80 GB)
        del model, pipe, latents, embeds, kt
        torch.cuda.empty_cache()
        out["vae"] = vae_section(cpu=not args.no_cpu_baseline, quick=args.quick_cpu_baseline)
    if rank == 0 and world == 1 and not args.no_vae and args.config == "c5" and not emu:
        # BASELINE config 5 = the denoise loop PLUS its one VAE encode of the masked conditioning video (predict_i2v.py:
        # pipeline_easyanimate_inpaint.py:1346-1383): timed here through the product pipeline method, reported beside the loop
        del model, pipe, latents, embeds, kt
        torch.cuda.empty_cache()
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import bench_vae
        from easyanimate_amd.pipeline import EasyAnimateInpaintPipeline, get_image_to_video_latent
        vae = bench_vae.build_vae()

        class _T:   # the two transformer attributes inpaint_conditioning reads
            resize_inpaint_mask_directly = True
            config = {"add_noise_in_inpaint_model": False}
        ip = EasyAnimateInpaintPipeline(vae=vae, transformer=None, scheduler=None)
        ip.transformer = _T()
        video, mask = get_image_to_video_latent(torch.rand(3, cfg["height"], cfg["width"]), cfg["frames"])
        with torch.no_grad():
            ip.inpaint_conditioning(video, mask, torch.bfloat16, device, True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            cond = ip.inpaint_conditioning(video, mask, torch.bfloat16, device, True)
            torch.cuda.synchronize()
        out["i2v_conditioning"] = {"seconds": time.perf_counter() - t0, "inpaint_latents_shape": list(cond.shape),
                                   "what": "host mask / masked-video construction + upload + one VAE encode of 49 frames + mask resize "
                                           "(once per pipeline call, not per step)"}
        del vae, ip
        torch.cuda.empty_cache()
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        t_block, kind, measured, threads, sample = cpu_baseline(S, (Fl, hl // 2, wl // 2), quick=args.quick_cpu_baseline)
        est_step_s = t_block * B * L
        out["cpu_baseline"] = {"value": 1.0 / est_step_s, "unit": "denoise-steps/s", "cores": threads, "kind": kind,
                               "block_at_full_S_measured": measured, "seconds_per_block_and_sample": t_block,
                               "sample": sample + f"; x B={B} x L={L} blocks = {est_step_s:.0f} s per denoise step (a full CPU step is "
                                                  f"{B * L} such blocks: hours -- the per-block time is measured, the product is arithmetic)"}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
