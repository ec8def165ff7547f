"""Out-of-bounds regression: the kernels with data-dependent addressing run once more behind the electric-fence
allocator of tools/guard_alloc (every tensor its own mapping, last byte = last mapped byte, unmapped pages on both sides, freed
ranges never reused), in a child process -- an over-read of 16 bytes is a GPU memory fault there, not a read of the caching
allocator's slack.  The positive control shows that the fence is armed on this box: a deliberate 4-byte read at the first byte
past a fenced tensor must kill its process with "Memory access fault", and the next process must start normally."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GUARDED = os.path.join(ROOT, "tools", "guarded.py")
SELECTION = ("permute_cols or swa_window_mapped or attention_window or segments or exchange_slot or kblocked_ffn_pair or "
             "gemm_w4a or qkv_fused_w4a or head_window or grouped_destination or "    # round 5: the hand-placed kernels and the head groups
             "eight_wave or layernorm_sweep")                                          # round 6: attention workgroup shapes / stages, LayerNorm sweep


def _run(args, env=None, timeout=600):
    e = dict(os.environ)
    e.update(env or {})
    return subprocess.run([sys.executable, GUARDED, *args], cwd=ROOT, env=e, capture_output=True, text=True, timeout=timeout)


@pytest.mark.parametrize("mode", ["right", "left"])
def test_kernels_behind_the_fence(mode):
    r = _run(["-m", "pytest", "tests/test_kernels_gpu.py", "-q", "-m", "gpu", "-p", "no:cacheprovider", "-x", "-k", SELECTION],
             env={"EA_GUARD_MODE": mode})
    tail = (r.stdout[-1500:] + r.stderr[-1500:])
    print(f"[fence] mode {mode}: rc {r.returncode}\n{tail}")
    assert r.returncode == 0, tail
    assert "passed" in r.stdout and "slack_hits=0" in r.stderr
    assert "Memory access fault" not in r.stderr and "OUT-OF-BOUNDS" not in r.stderr


def test_conv_w4a_behind_the_fence():
    """The four-wave row-slab convolutions (buffer-addressed slab requests, zero padding by the descriptor's range check)."""
    r = _run(["-m", "pytest", "tests/test_vae_gpu.py", "-q", "-m", "gpu", "-p", "no:cacheprovider", "-x", "-k",
              "conv_w4a or conv_on_blocked_input or groupnorm_blocked or tile_blend"])      # round 6: + the channel-blocked slab fill (64-bit base moves), tile blends
    tail = (r.stdout[-1500:] + r.stderr[-1500:])
    print(f"[fence] conv w4a: rc {r.returncode}\n{tail}")
    assert r.returncode == 0 and "passed" in r.stdout and "slack_hits=0" in r.stderr and "Memory access fault" not in r.stderr, tail


def test_text_encoder_kernels_behind_the_fence():
    """Round 6: the prompt-sized attention (operands straight from global memory, clamped rows), rotate-half scatter, SiLU-mul."""
    r = _run(["-m", "pytest", "tests/test_text_encoder_gpu.py", "-q", "-m", "gpu", "-p", "no:cacheprovider", "-x", "-k",
              "attention_causal_gqa or rope_half"])
    tail = (r.stdout[-1500:] + r.stderr[-1500:])
    print(f"[fence] text kernels: rc {r.returncode}\n{tail}")
    assert r.returncode == 0 and "passed" in r.stdout and "slack_hits=0" in r.stderr and "Memory access fault" not in r.stderr, tail


def test_fence_positive_control():
    r = _run(["--probe", "read", "0"], timeout=120)
    assert r.returncode != 0 and "Memory access fault" in r.stderr and "SURVIVED" not in r.stderr, r.stderr[-800:]
    ok = _run(["--probe", "read", "-4"], timeout=120)             # the last word INSIDE the tensor: must survive
    assert ok.returncode == 0 and "SURVIVED rc=0" in ok.stderr, ok.stderr[-800:]
