"""GPU parity of every HIP kernel, called through the C ABI (easyanimate_amd.ops -> ctypes), against a
plain PyTorch fp32/fp64 statement of the same op on the same bf16-rounded inputs.

Tolerances are stated per test; bf16 output rounding alone is 2^-9 relative (3.9e-3)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda"


def _needs_variants():
    """The superseded kernel generations (attention v1, four-wave GEMM, 32x32x16 row-slab conv) are compiled into
    EA_BUILD_VARIANTS=1 libraries only; their cross-check tests skip on the default library."""
    from easyanimate_amd import _lib
    if _lib.get_option("build_variants") != 1:
        pytest.skip("cross-check kernel generation: build with EA_BUILD_VARIANTS=1")


def _ops():
    from easyanimate_amd import ops
    return ops


def _report(name, got, ref):
    got = got.double().cpu()
    ref = ref.double().cpu()
    err = (got - ref).abs().max().item()
    rel = ((got - ref).norm() / (ref.norm() + 1e-30)).item()
    print(f"[parity] {name}: max_abs={err:.3e} rel_l2={rel:.3e} ref_absmax={ref.abs().max().item():.3e}")
    return err, rel


def _bf(x):
    return x.to(torch.bfloat16)


def _qkv_counters():
    """Launch counters of the fused QKV entry point; "gemm_qkv_fused_w4a" (set when the launch ran on the four-wave hand-placed
    kernel, the default for bf16 weights) is reported beside "gemm_qkv_fused", not instead of it."""
    from easyanimate_amd import _lib
    c = _lib.counters()
    c.pop("gemm_qkv_fused_w4a", None)
    return c


@pytest.mark.parametrize("B,R,D", [(2, 37, 3072), (1, 5, 128), (2, 300, 512), (1, 9, 8192)])
@pytest.mark.parametrize("affine,mod", [(True, True), (False, True), (True, False)])
def test_layernorm_modulate(B, R, D, affine, mod):
    ops = _ops()
    g = torch.Generator(device="cpu").manual_seed(1)
    x = _bf(torch.randn(B, R, D, generator=g) * 2 + 0.3).to(DEV)
    gamma = (1 + 0.1 * torch.randn(D, generator=g)).to(DEV) if affine else None
    beta = (0.1 * torch.randn(D, generator=g)).to(DEV) if affine else None
    table = torch.randn(B, 6 * D, generator=g).to(DEV)
    shift, scale = (table[:, 0:D], table[:, D:2 * D]) if mod else (None, None)
    y = ops.layernorm_modulate(x, gamma, beta, scale, shift, 1e-6)
    xr = x.double()
    ref = torch.nn.functional.layer_norm(xr, (D,), gamma.double() if affine else None,
                                         beta.double() if affine else None, 1e-6)
    if mod:
        ref = ref * (1 + scale.double()[:, None]) + shift.double()[:, None]
    err, rel = _report(f"layernorm_modulate {B}x{R}x{D}", y, ref)
    assert rel < 4e-3 and err < 0.08  # one bf16 rounding of values up to ~16


@pytest.mark.parametrize("B,R,D,wgs", [(2, 4099, 3072, 1024), (2, 4099, 3072, 7), (1, 3, 3072, 1024), (2, 513, 512, 16), (1, 260, 8192, 64)])
def test_layernorm_sweep_bit_identical_to_block_kernel(B, R, D, wgs):
    """Round 6: layernorm_modulate_sweep_kernel (workgroups sweep the tensor as one moving window, streaming loads / stores) does
    the round-1 kernel's operations on the same values in the same order per row: bit-identical for every grid size, streaming
    flag and ragged row count (rows % 4 != 0, fewer rows than waves, more trips than one)."""
    from easyanimate_amd import _lib
    ops = _ops()
    g = torch.Generator(device="cpu").manual_seed(5)
    x = _bf(torch.randn(B, R, D, generator=g) * 2 + 0.3).to(DEV)
    gamma, beta = (1 + 0.1 * torch.randn(D, generator=g)).to(DEV), (0.1 * torch.randn(D, generator=g)).to(DEV)
    table = torch.randn(B, 6 * D, generator=g).to(DEV)
    prev = _lib.get_option("ln_wgs"), _lib.get_option("ln_nt")
    try:
        _lib.set_option("ln_wgs", 0)
        ref = ops.layernorm_modulate(x, gamma, beta, table[:, D:2 * D], table[:, 0:D], 1e-6).clone()
        for nt in (0, 1, 2, 3):
            _lib.set_option("ln_wgs", wgs)
            _lib.set_option("ln_nt", nt)
            y = ops.layernorm_modulate(x, gamma, beta, table[:, D:2 * D], table[:, 0:D], 1e-6)
            assert torch.equal(y, ref), (nt, (y.float() - ref.float()).abs().max().item())
    finally:
        _lib.set_option("ln_wgs", prev[0])
        _lib.set_option("ln_nt", prev[1])


def test_rmsnorm():
    ops = _ops()
    g = torch.Generator(device="cpu").manual_seed(2)
    x = _bf(torch.randn(512, 3584, generator=g) * 30).to(DEV)
    w = (1 + 0.1 * torch.randn(3584, generator=g)).to(DEV)
    y = ops.rmsnorm(x, w, 1e-6)
    xr = x.double()
    ref = w.double() * (xr * torch.rsqrt(xr.pow(2).mean(-1, keepdim=True) + 1e-6))
    err, rel = _report("rmsnorm", y, ref)
    assert rel < 6e-3


@pytest.mark.parametrize("m,n,k,ai,ao", [(2, 18432, 512, 1, 0), (2, 512, 3072, 0, 1), (1, 100, 64, 0, 0), (4, 77, 512, 1, 1)])
def test_linear_small_m(m, n, k, ai, ao):
    ops = _ops()
    g = torch.Generator(device="cpu").manual_seed(3)
    x = torch.randn(m, k, generator=g).to(DEV)
    W = _bf(torch.randn(n, k, generator=g) / math.sqrt(k)).to(DEV)
    b = torch.randn(n, generator=g).to(DEV)
    y = ops.linear_small_m(x, W, b, ai, ao)
    xi = torch.nn.functional.silu(x.double()) if ai else x.double()
    ref = xi @ W.double().t() + b.double()
    if ao:
        ref = torch.nn.functional.silu(ref)
    err, rel = _report(f"linear_small_m {m}x{n}x{k}", y, ref)
    assert rel < 1e-5


def test_timestep_sinusoid():
    ops = _ops()
    t = torch.tensor([999.0, 500.5, 1.0], device=DEV)
    dim = 3072
    y = ops.timestep_sinusoid(t, dim, round_bf16=False)
    half = dim // 2
    f = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32, device=DEV) / half)
    a = t[:, None].float() * f[None]
    ref = torch.cat([torch.cos(a), torch.sin(a)], -1)
    err, rel = _report("timestep_sinusoid", y, ref)
    assert err < 2e-3  # fp32 sin/cos of arguments up to 1e3: argument rounding dominates
    yb = ops.timestep_sinusoid(t, dim, round_bf16=True)
    assert torch.equal(yb, yb.to(torch.bfloat16).float())


GEMM_SHAPES = [
    # B, M, N, K
    (1, 128, 128, 64),
    (1, 300, 192, 128),     # M and N tails
    (2, 257, 3072, 3072),   # batch, tail
    (1, 1000, 12288, 3072),
    (1, 512, 3072, 12288),
    (2, 64, 64, 3584),      # narrow N (proj_out-like), K = 56*64
    (1, 5, 8, 64),
    (1, 8300, 320, 128),    # tiles_m >= 64: per-XCD row bands with a short last band
    (2, 8192, 256, 64),
]


@pytest.mark.parametrize("B,M,N,K", GEMM_SHAPES)
@pytest.mark.parametrize("epi", [0, 1, 2])
def test_gemm(B, M, N, K, epi):
    ops = _ops()
    g = torch.Generator(device="cpu").manual_seed(4)
    # A is a strided view (row stride K+8, batch gap) to exercise lda / batch strides
    Abuf = _bf(torch.randn(B, M + 3, K + 8, generator=g)).to(DEV)
    A = Abuf[:, 1:M + 1, :K]
    W = _bf(torch.randn(N, K, generator=g) / math.sqrt(K)).to(DEV)
    bias = torch.randn(N, generator=g).to(DEV)
    res = _bf(torch.randn(B, M, N, generator=g)).to(DEV)
    gate = torch.randn(B, N, generator=g).to(DEV)
    ref = A.double() @ W.double().t() + bias.double()
    if epi == 1:
        ref = torch.nn.functional.gelu(ref, approximate="tanh")
    if epi == 2:
        ref = res.double() + gate.double()[:, None, :] * ref
        out = res.clone()  # in-place residual update, as the DiT block uses it
        y = ops.gemm(A, W, bias, epi, out=out, res=out, gate=gate)
    else:
        y = ops.gemm(A, W, bias, epi)
    err, rel = _report(f"gemm B{B} {M}x{N}x{K} epi{epi}", y, ref)
    assert rel < 4e-3, (err, rel)
    assert err < 0.06 * max(1.0, ref.abs().max().item() / 4)


GEMM256_SHAPES = [
    # B, M, N, K   (forced 256^2 ping-pong tile; odd/even K-tile counts, M and N tails, strided batches)
    (1, 256, 256, 64),
    (1, 256, 256, 128),
    (1, 512, 512, 192),
    (2, 700, 768, 3072),
    (1, 16500, 320, 320),    # tiles_m >= 64 (per-XCD bands), N tail (320 = 256 + 64)
    (2, 1000, 3072, 1024),
    (1, 2048, 12288, 256),
]


@pytest.mark.parametrize("B,M,N,K", GEMM256_SHAPES)
@pytest.mark.parametrize("epi", [0, 1, 2])
@pytest.mark.parametrize("mfma", [16, 32])
def test_gemm_tile256(B, M, N, K, epi, mfma):
    from easyanimate_amd import _lib
    _lib.set_option("gemm_tile", 256)
    _lib.set_option("gemm_mfma", mfma)
    try:
        test_gemm(B, M, N, K, epi)
    finally:
        _lib.set_option("gemm_tile", 0)
        _lib.set_option("gemm_mfma", 16)


def test_gemm_tile256_matches_tile128_bitwise_inputs_many_runs():
    """Race screen for the hand-placed vmcnt / barrier schedule: the 256^2 kernels must give the identical result on
    repeated launches at a DiT-sized problem.  The 32x32x16 version is bit-equal to the 128^2 kernel (same k order and
    fp32 accumulation chain per output element); the 16x16x32 version sums 32 k per instruction, so it agrees to fp32
    summation-order noise (a few last-bit bf16 flips)."""
    from easyanimate_amd import _lib
    ops = _ops()
    g = torch.Generator(device="cpu").manual_seed(11)
    M, N, K = 8192, 3072, 3072
    A = _bf(torch.randn(M, K, generator=g)).to(DEV)
    W = _bf(torch.randn(N, K, generator=g) / math.sqrt(K)).to(DEV)
    bias = torch.randn(N, generator=g).to(DEV)
    res = _bf(torch.randn(M, N, generator=g)).to(DEV)
    gate = torch.randn(1, N, generator=g).to(DEV)
    _lib.set_option("gemm_tile", 128)
    y128 = ops.gemm(A, W, bias, 0)
    y128g = ops.gemm(A, W, bias, 2, res=res, gate=gate)
    _lib.set_option("gemm_tile", 256)
    try:
        for mfma in (32, 16):
            _lib.set_option("gemm_mfma", mfma)
            y0 = ops.gemm(A, W, bias, 0)
            yg = ops.gemm(A, W, bias, 2, res=res, gate=gate)
            for _ in range(5):
                assert torch.equal(ops.gemm(A, W, bias, 0), y0)
                assert torch.equal(ops.gemm(A, W, bias, 2, res=res, gate=gate), yg)
            if mfma == 32:
                assert torch.equal(y0, y128) and torch.equal(yg, y128g)
            else:
                for a, b in ((y0, y128), (yg, y128g)):
                    d = (a.float() - b.float()).abs()
                    assert bool((d <= 2 ** -7 * b.float().abs().clamp_min(1.0)).all()) and (d > 0).float().mean().item() < 0.01
    finally:
        _lib.set_option("gemm_tile", 0)
        _lib.set_option("gemm_mfma", 16)


def test_gemm_identity_transpose_detect():
    """A = I with asymmetric W: catches swapped row/col in the C write (guide rule 16)."""
    ops = _ops()
    K = 128
    A = torch.eye(K, dtype=torch.bfloat16, device=DEV)
    W = _bf(torch.arange(256 * K, dtype=torch.float32).reshape(256, K) % 251 - 125).to(DEV)
    y = ops.gemm(A, W, None, 0)
    assert torch.equal(y.float(), W.float().t())


def _ref_qknorm_rope(qkv, H, nq_w, nq_b, nk_w, nk_b, cos, sin, eps):
    B, n, _ = qkv.shape
    q, k, v = qkv.float().chunk(3, dim=-1)

    def heads(t):
        return t.view(B, n, H, 64).transpose(1, 2)

    q, k, v = heads(q), heads(k), heads(v)
    q = torch.nn.functional.layer_norm(q, (64,), nq_w, nq_b, eps).to(torch.bfloat16)
    k = torch.nn.functional.layer_norm(k, (64,), nk_w, nk_b, eps).to(torch.bfloat16)

    def rope(x):
        if cos is None:
            return x
        xr, xi = x.reshape(*x.shape[:-1], -1, 2).unbind(-1)
        rot = torch.stack([-xi, xr], dim=-1).flatten(3)
        return (x.float() * cos[None, None] + rot.float() * sin[None, None]).to(torch.bfloat16)

    return rope(q), rope(k), v.to(torch.bfloat16)


@pytest.mark.parametrize("B,H,n_tok,seq_off,use_rope", [(2, 3, 200, 256, True), (1, 2, 77, 0, False), (2, 48, 130, 8, True), (1, 2, 64, 3, True)])
def test_qknorm_rope(B, H, n_tok, seq_off, use_rope):
    ops = _ops()
    g = torch.Generator(device="cpu").manual_seed(5)
    qkv = _bf(torch.randn(B, n_tok, 3 * H * 64, generator=g) * 1.5 + 0.2).to(DEV)
    nq_w, nk_w = [(1 + 0.2 * torch.randn(64, generator=g)).to(DEV) for _ in range(2)]
    nq_b, nk_b = [(0.2 * torch.randn(64, generator=g)).to(DEV) for _ in range(2)]
    ang = torch.rand(n_tok, 32, generator=g) * 6.28
    cos = ang.cos().repeat_interleave(2, 1).contiguous().to(DEV) if use_rope else None
    sin = ang.sin().repeat_interleave(2, 1).contiguous().to(DEV) if use_rope else None
    s_pad = ops.round_up(seq_off + n_tok, 256)
    q = torch.zeros(B, H, s_pad, 64, dtype=torch.bfloat16, device=DEV)
    k = torch.zeros_like(q)
    vt = torch.zeros(B, H, 64, s_pad, dtype=torch.bfloat16, device=DEV)
    ops.qknorm_rope(qkv, q, k, vt, nq_w, nq_b, nk_w, nk_b, cos, sin, seq_off, 1e-6)
    rq, rk, rv = _ref_qknorm_rope(qkv, H, nq_w, nq_b, nk_w, nk_b, cos, sin, 1e-6)
    # the folded softmax scale multiplies q only, ahead of the bf16 rounding
    q2, k2, vt2 = torch.zeros_like(q), torch.zeros_like(k), torch.zeros_like(vt)
    ops.qknorm_rope(qkv, q2, k2, vt2, nq_w, nq_b, nk_w, nk_b, cos, sin, seq_off, 1e-6, q_scale=ops.FOLDED_Q_SCALE)
    assert torch.equal(k2, k) and torch.equal(vt2, vt)
    e2, r2 = _report("qknorm_rope folded q", q2[:, :, seq_off:seq_off + n_tok], rq.float() * ops.FOLDED_Q_SCALE)
    assert r2 < 6e-3   # (the reference value here is the already rounded q times the scale: two roundings apart)
    sl = slice(seq_off, seq_off + n_tok)
    for name, got, ref in (("q", q[:, :, sl], rq), ("k", k[:, :, sl], rk), ("v", vt[:, :, :, sl].transpose(2, 3), rv)):
        err, rel = _report(f"qknorm_rope {name} B{B}H{H}n{n_tok}", got, ref)
        assert rel < 4e-3, name
    assert torch.equal(rv, vt[:, :, :, sl].transpose(2, 3))  # V is a pure copy/transposition
    # rows outside the written window stay zero
    assert q[:, :, :seq_off].abs().max().item() == 0 if seq_off else True
    assert q[:, :, seq_off + n_tok:].abs().max().item() == 0
    assert vt[:, :, :, seq_off + n_tok:].abs().max().item() == 0


@pytest.mark.parametrize("B,H,M,K,seq_off,use_rope", [(2, 4, 512, 128, 8, True), (1, 4, 256, 64, 0, False), (2, 48, 768, 3072, 256, True),
                                                     (1, 8, 1024, 192, 320, True),
                                                     # ragged last M tile: whole 8-token groups, a straddling group (M % 8 != 0),
                                                     # a last tile of one row, wave tiles entirely past M
                                                     (2, 4, 1032, 128, 8, True), (1, 8, 1283, 192, 256, True), (2, 4, 1025, 64, 0, False),
                                                     (1, 48, 2016, 3072, 256, True)])
def test_qkv_fused_matches_unfused(B, H, M, K, seq_off, use_rope):
    """ea_qkv_gemm_norm_rope_bf16 (one launch: three projections + qk-LayerNorm + RoPE + scatter in the GEMM epilogue,
    V^T through the operand-swapped main loop) against ea_gemm_bf16 x 3 (same 256^2 16x16x32 main loop) followed by
    ea_qknorm_rope_bf16: same roundings at the same points (V^T bit-identical; q / k up to the summation order of the
    LayerNorm statistics) -- and the unfused pieces are checked against fp64 elsewhere in this file."""
    from easyanimate_amd import _lib
    ops = _ops()
    g = torch.Generator(device="cpu").manual_seed(17)
    d = H * 64
    x = _bf(torch.randn(B, M, K, generator=g)).to(DEV)
    ws = [_bf(torch.randn(d, K, generator=g) / K ** 0.5).to(DEV) for _ in range(3)]
    bs = [(0.3 * torch.randn(d, generator=g)).to(DEV) for _ in range(3)]
    nq_w, nk_w = [(1 + 0.2 * torch.randn(64, generator=g)).to(DEV) for _ in range(2)]
    nq_b, nk_b = [(0.2 * torch.randn(64, generator=g)).to(DEV) for _ in range(2)]
    ang = torch.rand(M, 32, generator=g) * 6.28
    cos = ang.cos().repeat_interleave(2, 1).contiguous().to(DEV) if use_rope else None
    sin = ang.sin().repeat_interleave(2, 1).contiguous().to(DEV) if use_rope else None
    s_pad = ops.round_up(seq_off + M, 256) + 256
    def bufs():
        # poison: the fused kernel must write exactly rows / columns [seq_off, seq_off + M) and nothing else
        return (torch.full((B, H, s_pad, 64), 7.0, dtype=torch.bfloat16, device=DEV), torch.full((B, H, s_pad, 64), 7.0, dtype=torch.bfloat16, device=DEV),
                torch.full((B, H, 64, s_pad), 7.0, dtype=torch.bfloat16, device=DEV))
    q1, k1, vt1 = bufs()
    assert ops.qkv_fused_ok(M, d, K, seq_off)
    _lib.reset_counters()
    ops.qkv_gemm_norm_rope(x, ws[0], ws[1], ws[2], bs[0], bs[1], bs[2], q1, k1, vt1, nq_w, nq_b, nk_w, nk_b, cos, sin, seq_off,
                           1e-6, q_scale=ops.FOLDED_Q_SCALE)
    assert _qkv_counters() == {"gemm_qkv_fused": 1}
    q2, k2, vt2 = bufs()
    qkv = torch.empty(B, M, 3 * d, dtype=torch.bfloat16, device=DEV)
    _lib.set_option("gemm_tile", 256)
    try:
        for i in range(3):
            ops.gemm(x, ws[i], bs[i], ops.EPI_BIAS, out=qkv[:, :, i * d:(i + 1) * d])
    finally:
        _lib.set_option("gemm_tile", 0)
    ops.qknorm_rope(qkv, q2, k2, vt2, nq_w, nq_b, nk_w, nk_b, cos, sin, seq_off, 1e-6, q_scale=ops.FOLDED_Q_SCALE)
    torch.cuda.synchronize()
    # V: no reduction in its epilogue -> every bit agrees.  q / k: the 64-wide LayerNorm statistics are summed in a
    # different fp32 order (16 values per lane + 2 shuffles here, 8 + 3 there), so a few results land on the neighbouring
    # bf16 value: at most one ulp apart, and only a small fraction of them
    assert torch.equal(vt1, vt2)
    sl0 = slice(seq_off, seq_off + M)
    for name, a, b in (("q", q1[:, :, sl0], q2[:, :, sl0]), ("k", k1[:, :, sl0], k2[:, :, sl0])):
        af, bf_ = a.float(), b.float()
        diff = (af - bf_).abs()
        # one bf16 ulp of the row's largest value: a rotated pair x0*c - x1*s can cancel to ~0 while a last-bit change of
        # the normalised x0 still moves it by an ulp of x0
        tol = af.abs().amax(dim=-1, keepdim=True) * 2.0 ** -7 + 1e-30
        frac = (diff > 0).float().mean().item()
        print(f"[parity] fused qkv {name} B{B}H{H}M{M}K{K}: {frac * 100:.4f} % of the values differ from the unfused path, max "
              f"{(diff / tol).max().item():.2f} ulp of the row maximum")
        assert (diff <= tol).all() and frac < 0.02, name
    sl = slice(seq_off, seq_off + M)
    assert (q1[:, :, :seq_off] == 7).all() and (q1[:, :, seq_off + M:] == 7).all() and (vt1[:, :, :, seq_off + M:] == 7).all()
    # and against fp64 directly (loose: this is what the unfused tests pin tightly)
    ref_v = (x.double() @ ws[2].double().t() + bs[2].double())
    got_v = vt1[:, :, :, sl].permute(0, 3, 1, 2).reshape(B, M, d)
    err, rel = _report(f"fused qkv V B{B}H{H}M{M}K{K}", got_v, ref_v)
    assert rel < 4e-3
    # repeated launches are bit-identical (race screen)
    q3, k3, vt3 = bufs()
    ops.qkv_gemm_norm_rope(x, ws[0], ws[1], ws[2], bs[0], bs[1], bs[2], q3, k3, vt3, nq_w, nq_b, nk_w, nk_b, cos, sin, seq_off,
                           1e-6, q_scale=ops.FOLDED_Q_SCALE)
    assert torch.equal(q3, q1) and torch.equal(k3, k1) and torch.equal(vt3, vt1)


W8_SHAPES = [
    # B, M, N, K, tile: both kernels, M / N tails, odd / even K-tile counts, strided batch
    (1, 128, 128, 64, 128), (2, 300, 320, 192, 128), (1, 512, 3072, 3072, 128),
    (1, 256, 256, 64, 256), (1, 256, 512, 128, 256), (1, 512, 256, 192, 256), (2, 700, 768, 1024, 256), (1, 16500, 320, 320, 256),
    (1, 2048, 12288, 256, 256),
]


@pytest.mark.parametrize("B,M,N,K,tile", W8_SHAPES)
@pytest.mark.parametrize("epi", [0, 1, 2])
def test_gemm_fp8_weight_bit_identical(B, M, N, K, tile, epi):
    """ea_gemm_bf16_w8 (weight stored as float8_e4m3fn, widened inside the kernel; the storage mode of
    utils/fp8_optimization.py:17-35) against ea_gemm_bf16 on W.to(bfloat16) -- what the reference's per-call up-cast
    computes with: every bit agrees, for both tile sizes and every epilogue."""
    from easyanimate_amd import _lib
    ops = _ops()
    g = torch.Generator(device="cpu").manual_seed(29)
    A = _bf(torch.randn(B, M, K, generator=g)).to(DEV)
    W8 = (torch.randn(N, K, generator=g) / math.sqrt(K) * 4).to(torch.float8_e4m3fn).to(DEV)
    bias = torch.randn(N, generator=g).to(DEV)
    res = _bf(torch.randn(B, M, N, generator=g)).to(DEV) if epi == 2 else None
    gate = torch.randn(B, N, generator=g).to(DEV) if epi == 2 else None
    _lib.set_option("gemm_tile", tile)
    try:
        _lib.reset_counters()
        y8 = ops.gemm(A, W8, bias, epi, res=res, gate=gate)
        assert _lib.counters() == {("gemm_256_mi16_w8" if tile == 256 else "gemm_128_w8"): 1}
        yb = ops.gemm(A, W8.to(torch.bfloat16), bias, epi, res=res, gate=gate)
        assert torch.equal(y8, yb)
        for _ in range(3):                                   # race screen of the register-staged weight path
            assert torch.equal(ops.gemm(A, W8, bias, epi, res=res, gate=gate), y8)
    finally:
        _lib.set_option("gemm_tile", 0)
    ref = A.double() @ W8.to(torch.float32).double().t() + bias.double()
    if epi == 0:
        err, rel = _report(f"gemm fp8-weight B{B} {M}x{N}x{K} tile{tile}", y8, ref)
        assert rel < 4e-3


def test_fp8_widening_is_exact_for_every_code():
    """All 254 finite E4M3 codes through the in-kernel fp8 -> bf16 widening: one-hot activations pick single weights, so
    C[m, n] must equal float(W[n, m]) exactly (checks the OCP E4M3 decoding of v_cvt_pk_f32_fp8 on gfx950 against torch)."""
    ops = _ops()
    codes = torch.arange(256, dtype=torch.uint8)
    codes[codes == 0x7F] = 0
    codes[codes == 0xFF] = 0x80                                  # the two NaN codes: not weights
    W8 = codes.repeat(256 * 64 // 256).view(256, 64).contiguous()          # row n, column k: code (64 n + k) % 256
    W8 = W8.view(torch.float8_e4m3fn).to(DEV)
    A = torch.zeros(128, 64, dtype=torch.bfloat16)
    A[torch.arange(128), torch.arange(128) % 64] = 1.0
    for tile in (128, 256):
        from easyanimate_amd import _lib
        _lib.set_option("gemm_tile", tile)
        try:
            y = ops.gemm(A.to(DEV), W8, None, 0)                  # y[m, n] = W[n, m % 64]
        finally:
            _lib.set_option("gemm_tile", 0)
        want = W8.to(torch.float32).t()[torch.arange(128) % 64]   # [m, n]
        assert torch.equal(y.float(), want), tile


@pytest.mark.parametrize("B,H,M,K,seq_off", [(2, 4, 512, 128, 8), (1, 48, 768, 3072, 256)])
def test_qkv_fused_fp8_weight_bit_identical(B, H, M, K, seq_off):
    """ea_qkv_gemm_norm_rope_bf16_w8 against ea_qkv_gemm_norm_rope_bf16 on the up-cast weights."""
    from easyanimate_amd import _lib
    ops = _ops()
    g = torch.Generator(device="cpu").manual_seed(31)
    d = H * 64
    x = _bf(torch.randn(B, M, K, generator=g)).to(DEV)
    w8 = [(torch.randn(d, K, generator=g) / K ** 0.5 * 4).to(torch.float8_e4m3fn).to(DEV) for _ in range(3)]
    bs = [(0.3 * torch.randn(d, generator=g)).to(DEV) for _ in range(3)]
    nw = [(1 + 0.2 * torch.randn(64, generator=g)).to(DEV) for _ in range(2)]
    nb = [(0.2 * torch.randn(64, generator=g)).to(DEV) for _ in range(2)]
    ang = torch.rand(M, 32, generator=g) * 6.28
    cos, sin = ang.cos().repeat_interleave(2, 1).contiguous().to(DEV), ang.sin().repeat_interleave(2, 1).contiguous().to(DEV)
    s_pad = ops.round_up(seq_off + M, 256)
    outs = []
    for ws in (w8, [w.to(torch.bfloat16) for w in w8]):
        q, k = (torch.zeros(B, H, s_pad, 64, dtype=torch.bfloat16, device=DEV) for _ in range(2))
        vt = torch.zeros(B, H, 64, s_pad, dtype=torch.bfloat16, device=DEV)
        _lib.reset_counters()
        ops.qkv_gemm_norm_rope(x, ws[0], ws[1], ws[2], bs[0], bs[1], bs[2], q, k, vt, nw[0], nb[0], nw[1], nb[1], cos, sin, seq_off, 1e-6,
                               q_scale=ops.FOLDED_Q_SCALE)
        assert _qkv_counters() == {("gemm_qkv_fused_w8" if ws is w8 else "gemm_qkv_fused"): 1}
        outs.append((q, k, vt))
    for a, b in zip(*outs):
        assert torch.equal(a, b)


def _attn_inputs(B, H, S, seed, scale_q=1.0):
    ops = _ops()
    g = torch.Generator(device="cpu").manual_seed(seed)
    s_pad = ops.round_up(S, 256)
    q = torch.zeros(B, H, s_pad, 64, dtype=torch.bfloat16, device=DEV)
    k = torch.zeros_like(q)
    vt = torch.zeros(B, H, 64, s_pad, dtype=torch.bfloat16, device=DEV)
    q[:, :, :S] = _bf(torch.randn(B, H, S, 64, generator=g) * scale_q).to(DEV)
    k[:, :, :S] = _bf(torch.randn(B, H, S, 64, generator=g)).to(DEV)
    v = _bf(torch.randn(B, H, S, 64, generator=g)).to(DEV)
    vt[:, :, :, :S] = v.transpose(2, 3)
    return q, k, vt, v


def _attn_ref(q, k, v, S):
    qf, kf, vf = q[:, :, :S].double(), k[:, :, :S].double(), v.double()
    p = torch.softmax(qf @ kf.transpose(2, 3) / 8.0, dim=-1)
    o = p @ vf
    B, H = q.shape[:2]
    return o.transpose(1, 2).reshape(B, S, H * 64)


def _has_variants():
    from easyanimate_amd import _lib
    return _lib.get_option("build_variants") == 1


def _fold(q):
    """What ea_qknorm_rope_bf16(q_scale=FOLDED_Q_SCALE) hands the attention kernel: bf16(q * 64^-1/2 * log2 e)."""
    from easyanimate_amd import ops
    return (q.float() * ops.FOLDED_Q_SCALE).to(torch.bfloat16)


@pytest.fixture(params=[2, 3])
def attn_kernel(request):
    """Run a test with the 32x32x16 (v2) and the 16x16x32 (v3, default) pipelined attention kernels.  v3 serves calls
    with the scale folded into Q; everything else falls through to v2."""
    from easyanimate_amd import _lib
    if request.param != 3:
        _needs_variants()          # the default library carries v3 only (round 4)
    _lib.set_option("attn_variant", request.param)
    yield request.param
    _lib.set_option("attn_variant", 3)



@pytest.mark.parametrize("folded", [False, True])
@pytest.mark.parametrize("B,H,S", [(1, 1, 64), (1, 2, 256), (2, 3, 333), (1, 2, 1000), (2, 9, 2048 + 77), (1, 1, 5)])
def test_attention(B, H, S, folded, attn_kernel):
    """folded: the softmax scale lives in Q and the kernel runs its RAW path (P = exp2 of the raw scores, no shift)."""
    ops = _ops()
    q, k, vt, v = _attn_inputs(B, H, S, 7, scale_q=2.0)
    if folded:
        qs = _fold(q)
        out = ops.attention(qs, k, vt, S, ops.FOLDED_ATTN_SCALE)
        ref = _attn_ref((qs.float() / ops.FOLDED_Q_SCALE).double(), k, v, S)   # the reference of the rounded, folded q
    else:
        _needs_variants()          # un-folded scale: the v2 generation (cross-check libraries only)
        out = ops.attention(q, k, vt, S, 0.125)
        ref = _attn_ref(q, k, v, S)
    err, rel = _report(f"attention B{B}H{H}S{S} folded={folded}", out, ref)
    assert rel < 8e-3 and err < 0.05, (err, rel)


@pytest.mark.parametrize("B,H,S,window", [(1, 2, 1000, 64), (2, 3, 2304, 768), (1, 1, 300, 5), (1, 2, 520, 4000), (1, 1, 777, 0)])
def test_attention_window(B, H, S, window):
    """ea_attention_window_fwd_bf16 (the SWA processor's band attention, processor.py:420): |i - j| <= window, against an
    fp64 masked softmax.  Covers bands narrower than a key tile, wider than the sequence, and window 0 (self only)."""
    ops = _ops()
    q, k, vt, v = _attn_inputs(B, H, S, 21)
    out = ops.attention_window(q, k, vt, S, window, 0.125)
    qd, kd, vd = q[:, :, :S].double(), k[:, :, :S].double(), v.double()
    s = qd @ kd.transpose(2, 3) * 0.125
    i = torch.arange(S, device=s.device)
    s = s.masked_fill(((i[:, None] - i[None, :]).abs() > window)[None, None], float("-inf"))
    ref = (torch.softmax(s, -1) @ vd).transpose(1, 2).reshape(B, S, H * 64)
    err, rel = _report(f"attention window B{B}H{H}S{S}w{window}", out, ref)
    assert rel < 5e-3 and torch.isfinite(out.float()).all()


def test_attention_forced_rescale_and_padding_garbage():
    """Spike keys late in the sequence so the running max jumps in a late tile (online-softmax rescale branch),
    and poison the padded tail of q/k to prove masked keys cannot leak (vt tail must stay finite by contract)."""
    ops = _ops()
    B, H, S = 1, 2, 700
    q, k, vt, v = _attn_inputs(B, H, S, 11)
    k[:, :, 650] = q[:, :, 3] * 6  # huge score for query 3 at key 650 (tile 10)
    k[:, :, 130] = q[:, :, 300] * 5
    q[:, :, S:] = 1e30
    k[:, :, S:] = -1e30
    qs = _fold(q)
    out = ops.attention(qs, k, vt, S, ops.FOLDED_ATTN_SCALE)
    ref = _attn_ref((qs[:, :, :S].float() / ops.FOLDED_Q_SCALE).double(), k, v, S)
    err, rel = _report("attention forced-rescale", out, ref)
    assert torch.isfinite(out.float()).all()
    assert rel < 8e-3 and err < 0.05


@pytest.mark.parametrize("variant", [1, 2, 3])
def test_attention_extreme_dynamic_range(variant):
    """Scores spanning +-600 (log2 units) inside one query row, row maxima in either key half of a 32-key block:
    every exponential that is not the row maximum's under- or overflows.  (Caught a mis-compiled cross-half max in
    the rescale path: the running max came from the lower key half only, finite until the halves differ by 2^127.)"""
    from easyanimate_amd import _lib
    ops = _ops()
    if variant != 3:
        _needs_variants()
    _lib.set_option("attn_variant", variant)
    try:
        B, H, S = 1, 3, 1000
        q, k, vt, v = _attn_inputs(B, H, S, 37)
        q = (q.float() * 30.0).to(torch.bfloat16)
        for (qq, kk, a) in [(3, 40, 3.0), (3, 70, 2.5), (17, 31, 4.0), (200, 999, 9.0), (777, 960, 5.0), (64, 0, 6.0)]:
            k[:, 0, kk] = (q[:, 0, qq].float() / 30.0 * a).to(torch.bfloat16)
        if variant == 3:       # v3 takes the scale folded into Q: its RAW mode has to hand over to the shifted loop here
            qs = _fold(q)
            out = ops.attention(qs, k, vt, S, ops.FOLDED_ATTN_SCALE)
            ref = _attn_ref((qs.float() / ops.FOLDED_Q_SCALE).double(), k, v, S)
        else:
            out = ops.attention(q, k, vt, S, 0.125)
            ref = _attn_ref(q, k, v, S)
        assert torch.isfinite(out.float()).all()
        err, rel = _report(f"attention v{variant} extreme range", out, ref)
        assert rel < 8e-3 and err < 0.05
    finally:
        _lib.set_option("attn_variant", 3)


@pytest.mark.parametrize("amp", [0.02, 1.0, 30.0])
def test_attention_folded_leaves_raw_mode(amp, attn_kernel):
    """The RAW path (m = 0, P = exp2(raw score)) has to hand over to the shifted path when a row sum leaves
    [2^-60, 2^60): scores scaled to +-a few (stays RAW), to hundreds (overflow side, head 0 / underflow side, head 1:
    every key of a query far below zero until a late block), spikes in even / odd blocks and in the ragged tail."""
    ops = _ops()
    B, H, S = 1, 3, 1000
    q, k, vt, v = _attn_inputs(B, H, S, 37)
    q = (q.float() * amp).to(torch.bfloat16)
    for (qq, kk, a) in [(3, 40, 3.0), (3, 70, 2.5), (17, 31, 4.0), (200, 999, 9.0), (777, 960, 5.0), (64, 0, 6.0)]:
        k[:, 0, kk] = (q[:, 0, qq].float() / max(amp, 1e-3) * a).to(torch.bfloat16)
    # head 1: all scores strongly negative (q . k = -|const|) except a few late keys
    d = torch.ones(64, device=DEV)
    q[:, 1, :S] = (d * 2.0 * amp).to(torch.bfloat16)
    k[:, 1, :S] = (-d * 1.5 + 0.05 * torch.randn(S, 64, device=DEV)).to(torch.bfloat16)
    k[:, 1, 900:905] = (d * 0.5).to(torch.bfloat16)
    qs = _fold(q)
    out = ops.attention(qs, k, vt, S, ops.FOLDED_ATTN_SCALE)
    ref = _attn_ref((qs.float() / ops.FOLDED_Q_SCALE).double(), k, v, S)
    assert torch.isfinite(out.float()).all()
    for h in range(H):
        err, rel = _report(f"attention folded amp {amp} head{h}", out[:, :, h * 64:(h + 1) * 64], ref[:, :, h * 64:(h + 1) * 64])
        assert rel < 8e-3 and err < 0.05, (h, err, rel)


@pytest.mark.parametrize("variant", [1, 2, 3])
def test_attention_variants_rescale_paths(variant):
    """Both kernels (v1: per-block rescale; v2: software-pipelined, deferred rescale with the pending P.V flushed in the
    rare branch) against an fp64 reference on inputs that force the rescale branch at chosen blocks (CDNA4 guide,
    rule 26): spikes in even / odd 32-key blocks, in the first tile, in the ragged last tile, growth below and above
    the defer threshold, and a steady ramp whose cumulated growth crosses the threshold many times."""
    from easyanimate_amd import _lib
    ops = _ops()
    if variant != 3:
        _needs_variants()
    _lib.set_option("attn_variant", variant)
    try:
        B, H, S = 1, 3, 1000   # 15 full tiles + a 40-key tail
        q, k, vt, v = _attn_inputs(B, H, S, 23)
        # head 0: spikes of different heights at different block parities
        for (qq, kk, amp) in [(3, 40, 1.2), (3, 70, 2.5), (3, 650, 7.0), (17, 31, 4.0), (200, 96, 3.0), (200, 999, 9.0),
                              (777, 960, 5.0), (64, 0, 6.0), (65, 130, 0.6)]:
            k[:, 0, kk] = q[:, 0, qq] * amp
        # head 1: scores ramp up along the key axis for every query: k_j = (j/S) * 3 * mean-direction
        d = _bf(torch.ones(64)).to(DEV)
        q[:, 1, :S] = (q[:, 1, :S].float() * 0.3 + d.float() * 2.0).to(torch.bfloat16)
        ramp = (torch.arange(S, device=DEV).float() / S)[None, :, None]
        k[:, 1, :S] = (k[:, 1, :S].float() * 0.3 + d.float()[None, None, :] * 3.0 * ramp).to(torch.bfloat16)
        # head 2: plain random
        if variant == 3:
            qs = _fold(q)
            out = ops.attention(qs, k, vt, S, ops.FOLDED_ATTN_SCALE)
            ref = _attn_ref((qs[:, :, :S].float() / ops.FOLDED_Q_SCALE).double(), k, v, S)
        else:
            out = ops.attention(q, k, vt, S, 0.125)
            ref = _attn_ref(q, k, v, S)
        for h in range(H):
            err, rel = _report(f"attention v{variant} rescale-paths head{h}", out[:, :, h * 64:(h + 1) * 64],
                               ref[:, :, h * 64:(h + 1) * 64])
            assert rel < 8e-3 and err < 0.05, (h, err, rel)
        assert torch.isfinite(out.float()).all()
    finally:
        _lib.set_option("attn_variant", 3)


def test_attention_v2_matches_v1_large():
    """The two kernels agree on a long sequence (836 tiles worth of pipeline steady state is covered by the model
    tests; here 130 tiles, ragged tail, 16 heads so every XCD slot is used)."""
    from easyanimate_amd import _lib
    ops = _ops()
    _needs_variants()
    B, H, S = 1, 16, 8300
    q, k, vt, v = _attn_inputs(B, H, S, 29, scale_q=1.5)
    _lib.set_option("attn_variant", 1)
    o1 = ops.attention(q, k, vt, S, 0.125)
    ref = _attn_ref(q, k, v, S)
    e1, r1 = _report("attention v1 S8300", o1, ref)
    for var in (2,):
        _lib.set_option("attn_variant", var)
        o2 = ops.attention(q, k, vt, S, 0.125)
        o2b = ops.attention(q, k, vt, S, 0.125)
        assert torch.equal(o2, o2b)
        e2, r2 = _report(f"attention v{var} S8300", o2, ref)
        assert r2 < 8e-3 and e2 < 0.05 and r2 < 1.5 * r1 + 1e-4
    _lib.set_option("attn_variant", 3)


@pytest.mark.parametrize("splits", [(0, 320, 1000), (0, 64, 128, 1000), (0, 960, 1000), (0, 256, 576, 999)])
@pytest.mark.parametrize("folded", [False, True])
def test_attention_resumable_key_ranges(splits, folded, attn_kernel):
    """ea_attention_fwd_range_bf16: chaining key ranges through the fp32 state equals the one-shot attention (the
    sequence-parallel overlap path).  Ranges are visited OUT of order (softmax is order invariant); ragged last range.
    folded: the production path (scale in Q, RAW mode, v3 when selected); the spike at key 700 makes the rows of head 1
    leave RAW mode inside a late range, so the state carries a non-zero running maximum into the next launch."""
    ops = _ops()
    B, H = 2, 3
    S = splits[-1]
    q, k, vt, v = _attn_inputs(B, H, S, 31, scale_q=1.5)
    k[:, 1, 700] = q[:, 1, 5] * (40 if folded else 6)     # a spike inside a late range
    if folded:
        qs, sc = _fold(q), ops.FOLDED_ATTN_SCALE
        ref = _attn_ref((qs.float() / ops.FOLDED_Q_SCALE).double(), k, v, S)
    else:
        _needs_variants()
        qs, sc = q, 0.125
        ref = _attn_ref(q, k, v, S)
    qb, qe = 64, 900                   # a query sub-range, as a rank would own
    out = torch.full((B, S, H * 64), 3.0, dtype=torch.bfloat16, device=DEV)
    st = ops.attention_state(B, H, qb, qe, DEV)
    ranges = list(zip(splits[:-1], splits[1:]))
    order = ranges[1:] + ranges[:1]    # start with the second range, finish with the first
    for i, (lo, hi) in enumerate(order):
        ops.attention_range(qs, k, vt, sc, qb, qe, lo, hi, state=st, load_state=i > 0,
                            store_state=i < len(order) - 1, out=out)
    err, rel = _report(f"attention ranges {splits} folded={folded}", out[:, qb:qe], ref[:, qb:qe])
    assert torch.isfinite(out.float()).all()
    assert rel < 8e-3 and err < 0.05
    assert (out[:, :qb] == 3.0).all() and (out[:, qe:] == 3.0).all()
    one = ops.attention(qs, k, vt, S, sc, q_begin=qb, q_end=qe)
    assert (out[:, qb:qe].float() - one[:, qb:qe].float()).abs().max().item() < 0.02


@pytest.mark.parametrize("P,rank,nl,n_last", [(2, 0, 128, 128), (2, 1, 128, 70), (4, 1, 192, 192), (4, 3, 192, 100), (4, 0, 64, 5), (3, 2, 320, 17)])
def test_attention_segments_equals_contiguous_keys(P, rank, nl, n_last):
    """ea_attention_fwd_segments_bf16 (sequence parallelism: the remote K / V^T shards are read where all_gather_into_tensor
    left them, own segment skipped) against ea_attention_fwd_range_bf16 over the same keys copied into one contiguous
    layout -- same key order, same tiles, so every bit must agree; with and without a carried-in state."""
    ops = _ops()
    g = torch.Generator(device="cpu").manual_seed(31 + P + rank)
    B, H, T, n_own = 2, 3, 64, (n_last if rank == P - 1 else nl)
    S_q = T + n_own
    q_pad = ops.round_up(S_q, 256)
    q = torch.zeros(B, H, q_pad, 64, dtype=torch.bfloat16, device=DEV)
    q[:, :, :S_q] = _bf(torch.randn(B, H, S_q, 64, generator=g) * 0.3).to(DEV)
    gathered = torch.zeros(P, 2, B, H, nl * 64, dtype=torch.bfloat16, device=DEV)
    shard_rows = [nl] * (P - 1) + [n_last]
    for r in range(P):
        gathered[r, 0].view(B, H, nl, 64)[:, :, :shard_rows[r]] = _bf(torch.randn(B, H, shard_rows[r], 64, generator=g)).to(DEV)
        gathered[r, 1].view(B, H, 64, nl)[:, :, :, :shard_rows[r]] = _bf(torch.randn(B, H, 64, shard_rows[r], generator=g)).to(DEV)
    others = [r for r in range(P) if r != rank]
    kv_valid = sum(nl for r in others[:-1]) + shard_rows[others[-1]]
    # contiguous copy of the same keys, in the same order
    s_pad = ops.round_up(len(others) * nl, 256)
    k = torch.zeros(B, H, s_pad, 64, dtype=torch.bfloat16, device=DEV)
    vt = torch.zeros(B, H, 64, s_pad, dtype=torch.bfloat16, device=DEV)
    for i, r in enumerate(others):
        k[:, :, i * nl:(i + 1) * nl] = gathered[r, 0].view(B, H, nl, 64)
        vt[:, :, :, i * nl:(i + 1) * nl] = gathered[r, 1].view(B, H, 64, nl)
    kk = torch.zeros(B, H, max(s_pad, q_pad), 64, dtype=torch.bfloat16, device=DEV)
    vv = torch.zeros(B, H, 64, max(s_pad, q_pad), dtype=torch.bfloat16, device=DEV)
    qq = torch.zeros_like(kk)
    kk[:, :, :s_pad], vv[:, :, :, :s_pad], qq[:, :, :q_pad] = k, vt, q
    o_ref = torch.empty(B, S_q, H * 64, dtype=torch.bfloat16, device=DEV)
    ops.attention_range(qq, kk, vv, ops.FOLDED_ATTN_SCALE, 0, S_q, 0, kv_valid, out=o_ref)
    o_seg = torch.empty_like(o_ref)
    ops.attention_segments(q, gathered, P, rank, nl, kv_valid, 0, S_q, out=o_seg)
    assert torch.equal(o_seg, o_ref)
    # resumed from a stored state (the local-key pass of the sequence-parallel block)
    st_a = ops.attention_state(B, H, 0, S_q, DEV)
    st_b = ops.attention_state(B, H, 0, S_q, DEV)
    kl = _bf(torch.randn(B, H, 256, 64, generator=g)).to(DEV)
    kloc = torch.zeros_like(qq); vloc = torch.zeros_like(vv)
    kloc[:, :, :256] = kl
    vloc[:, :, :, :256] = _bf(torch.randn(B, H, 64, 256, generator=g)).to(DEV)
    ops.attention_range(qq, kloc, vloc, ops.FOLDED_ATTN_SCALE, 0, S_q, 0, 200, state=st_a, store_state=True)
    st_b.copy_(st_a)
    kq = torch.zeros(B, H, q_pad, 64, dtype=torch.bfloat16, device=DEV)
    o1, o2 = torch.empty_like(o_ref), torch.empty_like(o_ref)
    ops.attention_range(qq, kk, vv, ops.FOLDED_ATTN_SCALE, 0, S_q, 0, kv_valid, state=st_a, load_state=True, out=o1)
    ops.attention_segments(q, gathered, P, rank, nl, kv_valid, 0, S_q, state=st_b, load_state=True, out=o2)
    assert torch.equal(o1, o2) and not torch.equal(o1, o_ref)


@pytest.mark.parametrize("B,H,S,qb,qe", [(1, 3, 1000, 0, 1000), (2, 2, 2300, 64, 2100), (1, 9, 777, 100, 300), (1, 1, 5000, 0, 5000)])
def test_attention_eight_wave_workgroups_bit_identical(B, H, S, qb, qe):
    """Round 6: ea_set_option("attn_nw", 8) -- 512 queries per workgroup, one workgroup per CU, ONE K / V^T stream per CU (half the
    LDS-DMA requests per wave) -- runs every wave's instruction stream over the same 64 queries and the same key tiles as the
    four-wave shape: bit-identical outputs for the plain launch, for chained key ranges whose fp32 state is written by one shape and
    read by the other (the state keeps the 256-query geometry), and for the segment form; head counts that are no multiple of 8
    (the tail split over the XCDs) and query ranges that end inside a 512-row block included."""
    from easyanimate_amd import _lib
    ops = _ops()
    q, k, vt, v = _attn_inputs(B, H, S, 77 + S, scale_q=1.5)
    k[:, 0, S // 2] = q[:, 0, min(5, S - 1)] * 40                       # one head leaves RAW mode half-way
    qs, sc = _fold(q), ops.FOLDED_ATTN_SCALE
    outs = {}

    def shape(nw, stages):
        _lib.set_option("attn_nw", nw)
        _lib.set_option("attn_stages", stages)
    SHAPES = ((4, 2), (8, 2), (4, 3))          # (waves per workgroup, LDS stages): the product shape, one stream per CU, a deeper ring
    try:
        for sh in SHAPES:
            shape(*sh)
            o = torch.full((B, S, H * 64), 3.0, dtype=torch.bfloat16, device=DEV)
            ops.attention(qs, k, vt, S, sc, out=o, q_begin=qb, q_end=qe)
            outs[sh] = o
        assert all(torch.equal(outs[(4, 2)], o) for o in outs.values()) and torch.isfinite(outs[(4, 2)][:, qb:qe].float()).all()
        # key ranges: the state written with one shape, resumed with another
        mid = (S // 2) // 64 * 64
        res = {}
        for first in SHAPES:
            for second in SHAPES:
                st = ops.attention_state(B, H, qb, qe, DEV)
                o = torch.full((B, S, H * 64), 3.0, dtype=torch.bfloat16, device=DEV)
                shape(*first)
                ops.attention_range(qs, k, vt, sc, qb, qe, mid, S, state=st, store_state=True)
                shape(*second)
                ops.attention_range(qs, k, vt, sc, qb, qe, 0, mid, state=st, load_state=True, out=o)
                res[(first, second)] = o
        assert all(torch.equal(res[((4, 2), (4, 2))], r) for r in res.values())
        assert (res[((4, 2), (4, 2))][:, qb:qe].float() - outs[(4, 2)][:, qb:qe].float()).abs().max().item() < 0.03
    finally:
        shape(4, 2)


@pytest.mark.parametrize("P,rank,nl,n_last", [(4, 1, 192, 192), (4, 3, 192, 100), (3, 2, 320, 17)])
def test_attention_segments_eight_wave_workgroups_bit_identical(P, rank, nl, n_last):
    from easyanimate_amd import _lib
    ops = _ops()
    g = torch.Generator(device="cpu").manual_seed(131 + P + rank)
    B, H, T, n_own = 2, 3, 64, (n_last if rank == P - 1 else nl)
    S_q = T + n_own
    q_pad = ops.round_up(S_q, 256)
    q = torch.zeros(B, H, q_pad, 64, dtype=torch.bfloat16, device=DEV)
    q[:, :, :S_q] = _bf(torch.randn(B, H, S_q, 64, generator=g) * 0.3).to(DEV)
    gathered = torch.zeros(P, 2, B, H, nl * 64, dtype=torch.bfloat16, device=DEV)
    shard_rows = [nl] * (P - 1) + [n_last]
    for r in range(P):
        gathered[r, 0].view(B, H, nl, 64)[:, :, :shard_rows[r]] = _bf(torch.randn(B, H, shard_rows[r], 64, generator=g)).to(DEV)
        gathered[r, 1].view(B, H, 64, nl)[:, :, :, :shard_rows[r]] = _bf(torch.randn(B, H, 64, shard_rows[r], generator=g)).to(DEV)
    others = [r for r in range(P) if r != rank]
    kv_valid = sum(nl for r in others[:-1]) + shard_rows[others[-1]]
    outs = {}
    try:
        for nw, stages in ((4, 2), (8, 2), (4, 3)):
            _lib.set_option("attn_nw", nw)
            _lib.set_option("attn_stages", stages)
            o = torch.empty(B, S_q, H * 64, dtype=torch.bfloat16, device=DEV)
            ops.attention_segments(q, gathered, P, rank, nl, kv_valid, 0, S_q, out=o)
            outs[(nw, stages)] = o
    finally:
        _lib.set_option("attn_nw", 4)
        _lib.set_option("attn_stages", 2)
    assert all(torch.equal(outs[(4, 2)], o) for o in outs.values())


@pytest.mark.parametrize("B,H,S", [(1, 3, 1024), (2, 2, 2300), (1, 9, 777), (1, 1, 5003)])
def test_attention_ping_pong_experiment_kernel(B, H, S):
    """attention_fwd_v5_kernel (ea_set_option("attn_nw", 16) in an EA_BUILD_VARIANTS=1 library; a measurement kernel, DESIGN 3.1:
    15 % slower than v3, profiles/r06l_attention_pingpong_ab.jsonl): the v3 arithmetic with the MFMA and
    softmax phases of the two waves of a SIMD run one phase apart.  Same products and exponentials; the row sums are fp32 adds of the
    unrounded weights instead of an MFMA over the rounded ones -> equal to v3 to bf16 noise, and to the fp64 reference inside the
    usual tolerance; ragged last tiles, head counts that are no multiple of 8, query counts that are no multiple of 512."""
    from easyanimate_amd import _lib
    _needs_variants()
    ops = _ops()
    q, k, vt, v = _attn_inputs(B, H, S, 177 + S, scale_q=1.5)
    qs, sc = _fold(q), ops.FOLDED_ATTN_SCALE
    ref = _attn_ref((qs.float() / ops.FOLDED_Q_SCALE).double(), k, v, S)
    try:
        _lib.set_option("attn_nw", 4)
        o3 = ops.attention(qs, k, vt, S, sc)
        _lib.set_option("attn_nw", 16)
        o5 = ops.attention(qs, k, vt, S, sc)
        o5b = ops.attention(qs, k, vt, S, sc)
    finally:
        _lib.set_option("attn_nw", 4)
    err, rel = _report(f"attention v5 (ping-pong) B={B} H={H} S={S}", o5, ref)
    assert torch.isfinite(o5.float()).all() and torch.equal(o5, o5b)
    assert rel < 8e-3 and err < 0.05 and (o5.float() - o3.float()).abs().max().item() < 0.03


def test_attention_full_size_config3():
    """BASELINE.json config 3 sequence (S = 53 504 = 836 key tiles, 209 query blocks), 2 heads: the product kernel against
    a chunked fp32 torch evaluation of the same attention on the GPU (checker only), plus the size-independent
    property the domain offers -- softmax attention is invariant to a permutation of the (key, value) rows."""
    ops = _ops()
    B, H, S = 1, 2, 53504
    q, k, vt, v = _attn_inputs(B, H, S, 41, scale_q=1.5)
    qs = _fold(q)
    out = ops.attention(qs, k, vt, S, ops.FOLDED_ATTN_SCALE)
    qf = qs[:, :, :S].float() / ops.FOLDED_Q_SCALE
    ref = torch.empty(B, S, H * 64, dtype=torch.float32, device=DEV)
    kf, vf = k[:, :, :S].float(), v.float()
    for lo in range(0, S, 4096):
        p = torch.softmax(qf[:, :, lo:lo + 4096] @ kf.transpose(2, 3) / 8.0, dim=-1)
        ref[:, lo:lo + 4096] = (p @ vf).transpose(1, 2).reshape(B, -1, H * 64)
    err, rel = _report("attention full size S=53504", out, ref)
    assert rel < 8e-3 and err < 0.02
    perm = torch.randperm(S, device=DEV)
    k2, vt2 = torch.zeros_like(k), torch.zeros_like(vt)
    k2[:, :, :S] = k[:, :, perm]
    vt2[:, :, :, :S] = vt[:, :, :, perm]
    out2 = ops.attention(qs, k2, vt2, S, ops.FOLDED_ATTN_SCALE)
    e2, r2 = _report("attention full size, keys permuted vs not", out2, out.float())
    assert r2 < 6e-3


def test_gemm_full_size_config3_sampled_rows():
    """The three DiT GEMM shapes at config-3 size (M = 106 496 rows): 512 sampled output rows against fp64."""
    ops = _ops()
    g = torch.Generator(device="cpu").manual_seed(43)
    M = 106496
    rows = torch.randint(0, M, (512,), generator=g).to(DEV)
    for (N, K, epi) in ((3072, 3072, 0), (12288, 3072, 1), (3072, 12288, 2)):
        A = torch.randn(M, K, device=DEV).to(torch.bfloat16)
        W = (torch.randn(N, K, device=DEV) / math.sqrt(K)).to(torch.bfloat16)
        bias = torch.randn(N, device=DEV)
        gate = torch.randn(1, N, device=DEV)
        res = torch.randn(M, N, device=DEV).to(torch.bfloat16)
        ref = A[rows].double() @ W.double().t() + bias.double()
        if epi == 1:
            ref = torch.nn.functional.gelu(ref, approximate="tanh")
        if epi == 2:
            ref = res[rows].double() + gate.double() * ref
            y = ops.gemm(A, W, bias, 2, out=res, res=res, gate=gate)
        else:
            y = ops.gemm(A, W, bias, epi)
        err, rel = _report(f"gemm full size {M}x{N}x{K} epi{epi} (512 rows)", y[rows], ref)
        assert rel < 4e-3
        del A, W, res, y


def test_attention_query_range(attn_kernel):
    """Sequence-parallel use: only rows [q_begin, q_end) are produced, the rest of `out` is untouched (plain and
    folded scale: v2 / v3)."""
    ops = _ops()
    B, H, S = 1, 2, 1300
    q, k, vt, v = _attn_inputs(B, H, S, 13)
    cases = [(_fold(q), ops.FOLDED_ATTN_SCALE, _attn_ref((_fold(q).float() / ops.FOLDED_Q_SCALE).double(), k, v, S))]
    if _has_variants():
        cases.append((q, 0.125, _attn_ref(q, k, v, S)))
    for qs, sc, ref in cases:
        out = torch.full((B, S, H * 64), 7.0, dtype=torch.bfloat16, device=DEV)
        ops.attention(qs, k, vt, S, sc, out=out, q_begin=512, q_end=1024)
        err, rel = _report("attention q-range", out[:, 512:1024], ref[:, 512:1024])
        assert rel < 8e-3
        assert (out[:, :512] == 7.0).all() and (out[:, 1024:] == 7.0).all()


def test_patchify_unpatchify_cfg_euler():
    ops = _ops()
    g = torch.Generator(device="cpu").manual_seed(17)
    B, C, F, H, W = 2, 16, 3, 8, 12
    lat = torch.randn(B, C, F, H, W, generator=g).to(DEV)
    extra = torch.randn(B, 17, F, H, W, generator=g).to(DEV)
    for ex, kp in ((None, 64), (extra, 192)):
        for dt in (torch.float32, torch.bfloat16):
            l = lat.to(dt)
            e = ex.to(dt) if ex is not None else None
            cols = ops.patchify(l, e, kp)
            x = l if e is None else torch.cat([l, e], 1)
            Ct = x.shape[1]
            ref = x.reshape(B, Ct, F, H // 2, 2, W // 2, 2).permute(0, 2, 3, 5, 1, 4, 6).reshape(B, F * (H // 2) * (W // 2), Ct * 4)
            assert torch.equal(cols[:, :, :Ct * 4], ref.to(torch.bfloat16))
            assert cols[:, :, Ct * 4:].abs().max().item() == 0 if kp > Ct * 4 else True
    tok = _bf(torch.randn(B, F * (H // 2) * (W // 2), C * 4, generator=g)).to(DEV)
    for dt in (torch.float32, torch.bfloat16):
        out = ops.unpatchify(tok, C, F, H // 2, W // 2, dt)
        ref = tok.reshape(B, F, H // 2, W // 2, C, 2, 2).permute(0, 4, 1, 2, 5, 3, 6).flatten(5, 6).flatten(3, 4)
        assert torch.equal(out, ref.to(dt))
    for dt in (torch.float32, torch.bfloat16):
        v = torch.randn(2, C, F, H, W, generator=g).to(DEV).to(dt)
        x = torch.randn(1, C, F, H, W, generator=g).to(DEV).to(dt)
        x0 = x.clone()
        ops.cfg_euler_step(v, x, 6.0, -0.02, True)
        vv = (v[0:1] + 6.0 * (v[1:2] - v[0:1]))
        ref = (x0.float() + (-0.02) * vv.float()).to(dt)
        err, rel = _report(f"cfg_euler {dt}", x, ref)
        assert rel < (1e-6 if dt == torch.float32 else 6e-3)


def test_errors_are_reported_not_fatal():
    ops = _ops()
    A = torch.zeros(4, 100, dtype=torch.bfloat16, device=DEV)  # K=100 is not a multiple of 64
    W = torch.zeros(8, 100, dtype=torch.bfloat16, device=DEV)
    with pytest.raises(RuntimeError, match="multiple of 64"):
        ops.gemm(A, W, None, 0)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.gemm(A.cpu(), W.cpu(), None, 0)


# ---- round 3: K / V^T land in the rank's slot of the exchange buffer; K | V first; slot-relative key ranges ------------------
@pytest.mark.parametrize("fused,M", [(True, 512), (True, 1283), (True, 13104), (False, 200), (False, 1283)])
def test_qkv_projection_into_an_exchange_slot(fused, M):
    """ea_qkv_gemm_norm_rope_bf16 / ea_qknorm_rope_bf16 with their own K / V^T geometry (kv_off, kv_rows): q goes to the
    workspace rows [seq_off, ..), K / V^T to rows [kv_off, ..) of a buffer with another number of rows per head -- bit-identical
    to the same launch into buffers of Q's geometry, nothing else written; the fused launch split into parts (K | V, then Q)
    writes the same bits as the single launch."""
    from easyanimate_amd import _lib
    ops = _ops()
    g = torch.Generator(device="cpu").manual_seed(23)
    B, H, K = 2, 4, 256
    d = H * 64
    x = _bf(torch.randn(B, M, K, generator=g)).to(DEV)
    ws = [_bf(torch.randn(d, K, generator=g) / K ** 0.5).to(DEV) for _ in range(3)]
    bs = [(0.3 * torch.randn(d, generator=g)).to(DEV) for _ in range(3)]
    nq_w, nk_w = [(1 + 0.2 * torch.randn(64, generator=g)).to(DEV) for _ in range(2)]
    nq_b, nk_b = [(0.2 * torch.randn(64, generator=g)).to(DEV) for _ in range(2)]
    ang = torch.rand(M, 32, generator=g) * 6.28
    cos, sin = ang.cos().repeat_interleave(2, 1).contiguous().to(DEV), ang.sin().repeat_interleave(2, 1).contiguous().to(DEV)
    seq_off, kv_off = 64, 128
    # kv_rows as the product chooses it (sequence_parallel.Layout.rows: a multiple of 256); M = 1283 / 13104 end in a ragged 256-row
    # tile (round 3 ended on a red M = 1283 case whose kv_rows = kv_off + M + 64 was no multiple of 8: the entry point's own check)
    s_pad, kv_rows = ops.round_up(seq_off + M, 256), ops.round_up(kv_off + M + 64, 256)
    full = lambda *shape: torch.full(shape, 7.0, dtype=torch.bfloat16, device=DEV)

    def run(q, k, vt, **kw):
        if fused:
            assert ops.qkv_fused_ok(M, d, K, seq_off, kw.get("kv_off"))
            ops.qkv_gemm_norm_rope(x, ws[0], ws[1], ws[2], bs[0], bs[1], bs[2], q, k, vt, nq_w, nq_b, nk_w, nk_b, cos, sin, seq_off, 1e-6,
                                   q_scale=ops.FOLDED_Q_SCALE, **kw)
        else:
            assert "parts" not in kw
            qkv = torch.empty(B, M, 3 * d, dtype=torch.bfloat16, device=DEV)
            for i in range(3):
                ops.gemm(x, ws[i], bs[i], ops.EPI_BIAS, out=qkv[:, :, i * d:(i + 1) * d])
            ops.qknorm_rope(qkv, q, k, vt, nq_w, nq_b, nk_w, nk_b, cos, sin, seq_off, 1e-6, q_scale=ops.FOLDED_Q_SCALE, **kw)
    q0, k0, vt0 = full(B, H, s_pad, 64), full(B, H, s_pad, 64), full(B, H, 64, s_pad)
    run(q0, k0, vt0)
    q1, k1, vt1 = full(B, H, s_pad, 64), full(B, H, kv_rows, 64), full(B, H, 64, kv_rows)
    run(q1, k1, vt1, kv_off=kv_off)
    torch.cuda.synchronize()
    assert torch.equal(q1, q0)
    assert torch.equal(k1[:, :, kv_off:kv_off + M], k0[:, :, seq_off:seq_off + M]) and torch.equal(vt1[:, :, :, kv_off:kv_off + M], vt0[:, :, :, seq_off:seq_off + M])
    assert (k1[:, :, :kv_off] == 7).all() and (k1[:, :, kv_off + M:] == 7).all() and (vt1[:, :, :, :kv_off] == 7).all() and (vt1[:, :, :, kv_off + M:] == 7).all()
    if fused:
        q2, k2, vt2 = full(B, H, s_pad, 64), full(B, H, kv_rows, 64), full(B, H, 64, kv_rows)
        _lib.reset_counters()
        run(q2, k2, vt2, kv_off=kv_off, parts=ops.QKV_KV)
        torch.cuda.synchronize()
        assert (q2 == 7).all() and torch.equal(k2, k1) and torch.equal(vt2, vt1)      # K | V first: q untouched
        run(q2, k2, vt2, kv_off=kv_off, parts=ops.QKV_Q)
        torch.cuda.synchronize()
        assert torch.equal(q2, q1) and torch.equal(k2, k1) and torch.equal(vt2, vt1)
        assert _qkv_counters() == {"gemm_qkv_fused": 2, "gemm_qkv_fused_kv_part": 1, "gemm_qkv_fused_q_part": 1}


def test_qkv_exchange_slot_geometry_is_checked():
    """What round 3's red case actually hit: K / V^T rows per head that are no multiple of 8 (the V^T store moves 16-byte groups)
    are refused by the entry point -- an error string, not a wrong store."""
    ops = _ops()
    B, H, M, K, kv_off = 1, 4, 1283, 256, 128
    d, kv_rows = H * 64, 128 + 1283 + 64          # 1475: the geometry of the removed test case
    x = torch.zeros(B, M, K, dtype=torch.bfloat16, device=DEV)
    w, b = torch.zeros(d, K, dtype=torch.bfloat16, device=DEV), torch.zeros(d, device=DEV)
    n = torch.ones(64, device=DEV)
    cs = torch.zeros(M, 64, device=DEV)
    q = torch.zeros(B, H, ops.round_up(M, 256), 64, dtype=torch.bfloat16, device=DEV)
    k, vt = torch.zeros(B, H, kv_rows, 64, dtype=torch.bfloat16, device=DEV), torch.zeros(B, H, 64, kv_rows, dtype=torch.bfloat16, device=DEV)
    with pytest.raises(RuntimeError, match="multiples of 8"):
        ops.qkv_gemm_norm_rope(x, w, w, w, b, b, b, q, k, vt, n, n, n, n, cs, cs, 0, 1e-6, kv_off=kv_off)


@pytest.mark.parametrize("P,rank,T,nl,n_last", [(4, 1, 64, 192, 192), (4, 3, 128, 192, 100), (2, 0, 64, 128, 70), (3, 0, 192, 64, 64)])
def test_attention_segments_slot_rows(P, rank, T, nl, n_last):
    """The slot layout of the sequence-parallel exchange buffer: every segment is [text rows 0..T | shard rows T..T+nl];
    the own-slot pass (n_seg = 1 at the own slot, first_row = 0) attends text + own shard, the remote pass skips the own
    slot and the text rows of the others (first_row = T, used_rows = nl).  Both against ea_attention_fwd_range_bf16 over
    the same keys copied into one contiguous layout -- same key order, same tiles: bit-identical, state carried through."""
    ops = _ops()
    g = torch.Generator(device="cpu").manual_seed(77 + P + rank)
    B, H = 2, 3
    rows = T + nl
    shard_rows = [nl] * (P - 1) + [n_last]
    n_own = shard_rows[rank]
    S_q = T + n_own
    q_pad = ops.round_up(rows, 256)
    q = torch.zeros(B, H, q_pad, 64, dtype=torch.bfloat16, device=DEV)
    q[:, :, :S_q] = _bf(torch.randn(B, H, S_q, 64, generator=g) * 0.3).to(DEV)
    buf = torch.zeros(P, 2, B, H, rows * 64, dtype=torch.bfloat16, device=DEV)
    for r in range(P):
        n = T + shard_rows[r]       # every slot carries (its own copy of) the text rows: remote copies must never be read
        buf[r, 0].view(B, H, rows, 64)[:, :, :n] = _bf(torch.randn(B, H, n, 64, generator=g)).to(DEV)
        buf[r, 1].view(B, H, 64, rows)[:, :, :, :n] = _bf(torch.randn(B, H, 64, n, generator=g)).to(DEV)
    others = [r for r in range(P) if r != rank]
    remote_valid = sum(nl for r in others[:-1]) + shard_rows[others[-1]]
    # contiguous copy: [own slot rows 0..S_q | pad to 64 | remote shard rows ...]
    own_pad = ops.round_up(S_q, 64)
    s_pad = ops.round_up(max(own_pad + len(others) * nl, q_pad), 256)
    kk = torch.zeros(B, H, s_pad, 64, dtype=torch.bfloat16, device=DEV)
    vv = torch.zeros(B, H, 64, s_pad, dtype=torch.bfloat16, device=DEV)
    qq = torch.zeros_like(kk)
    qq[:, :, :q_pad] = q
    kk[:, :, :S_q] = buf[rank, 0].view(B, H, rows, 64)[:, :, :S_q]
    vv[:, :, :, :S_q] = buf[rank, 1].view(B, H, 64, rows)[:, :, :, :S_q]
    for i, r in enumerate(others):
        kk[:, :, own_pad + i * nl:own_pad + (i + 1) * nl] = buf[r, 0].view(B, H, rows, 64)[:, :, T:]
        vv[:, :, :, own_pad + i * nl:own_pad + (i + 1) * nl] = buf[r, 1].view(B, H, 64, rows)[:, :, :, T:]
    st_a, st_b = ops.attention_state(B, H, 0, S_q, DEV), ops.attention_state(B, H, 0, S_q, DEV)
    o_ref, o_seg = (torch.empty(B, S_q, H * 64, dtype=torch.bfloat16, device=DEV) for _ in range(2))
    ops.attention_range(qq, kk, vv, ops.FOLDED_ATTN_SCALE, 0, S_q, 0, S_q, state=st_a, store_state=True)
    ops.attention_range(qq, kk, vv, ops.FOLDED_ATTN_SCALE, 0, S_q, own_pad, own_pad + remote_valid, state=st_a, load_state=True, out=o_ref)
    ops.attention_segments(q, buf[rank], 1, -1, rows, S_q, 0, S_q, state=st_b, store_state=True, first_row=0, used_rows=ops.round_up(S_q, 64))
    ops.attention_segments(q, buf, P, rank, rows, remote_valid, 0, S_q, state=st_b, load_state=True, out=o_seg, first_row=T, used_rows=nl)
    torch.cuda.synchronize()
    assert torch.equal(o_seg, o_ref)
    # against fp64 softmax over exactly those keys (the text rows of the remote slots excluded)
    keys = torch.cat([kk[:, :, :S_q], kk[:, :, own_pad:own_pad + remote_valid]], 2).double()
    vals = torch.cat([vv[:, :, :, :S_q], vv[:, :, :, own_pad:own_pad + remote_valid]], 3).double().transpose(2, 3)
    qd = q[:, :, :S_q].double() / ops.FOLDED_Q_SCALE
    ref = (torch.softmax(qd @ keys.transpose(2, 3) / 8.0, -1) @ vals).transpose(1, 2).reshape(B, S_q, H * 64)
    err, rel = _report(f"segment attention over exchange slots P{P} r{rank}", o_seg, ref)
    assert rel < 8e-3





# ---- round 4: the SWA window pass without index copies ---------------------------------------------------------------------------
@pytest.mark.parametrize("grid", [(3, 70, 45), (2, 64, 64), (13, 8, 33), (1, 5, 7)])
def test_permute_cols_six_scan_orders(grid):
    """ea_permute_cols_bf16: V^T re-ordered along its token axis by each of the six axis orders of the (f, h, w) grid, against a
    torch gather with the reference's own permutation (base.permute(*order).reshape(-1), processor.py:400-417); tile tails in
    every axis (w not a multiple of 32, the destination's fastest axis not a multiple of 64), a column offset (the text rows in
    front), nothing written behind the sequence."""
    ops = _ops()
    F_, Hh, Ww = grid
    N = F_ * Hh * Ww
    B, H, T = 2, 7, 40
    orders = ((0, 1, 2), (0, 2, 1), (1, 0, 2), (1, 2, 0), (2, 0, 1), (2, 1, 0))
    g = torch.Generator(device="cpu").manual_seed(3)
    src_pad, dst_pad = ops.round_up(T + N, 256), ops.round_up(N, 64)
    src = torch.randn(B, H, 64, src_pad, generator=g).to(torch.bfloat16).to(DEV)
    dst = torch.full((B, H, 64, dst_pad), 7.0, dtype=torch.bfloat16, device=DEV)
    head_order = torch.tensor([h % 6 for h in range(H)], dtype=torch.int32, device=DEV)
    ops.permute_cols(src, dst, head_order, grid, T)
    torch.cuda.synchronize()
    base = torch.arange(N, device=DEV).view(F_, Hh, Ww)
    for h in range(H):
        tok = base.permute(*orders[h % 6]).reshape(-1)
        assert torch.equal(dst[:, h, :, :N], src[:, h, :, T + tok]), (h, orders[h % 6])
    assert (dst[:, :, :, N:] == 7).all()


@pytest.mark.parametrize("grid,T,H,cross_size", [((3, 16, 24), 64, 12, 256), ((2, 20, 13), 7, 6, 128), ((5, 8, 8), 40, 7, 96)])
def test_swa_window_mapped_equals_index_copies(grid, T, H, cross_size):
    """The SWA attend with the mapped window kernel (q / k addressed through the scan-order map, V^T permuted by tiled transposes,
    token-order store with the cross pass added) against its first version (torch index copies into scan order,
    ea_attention_window_fwd_bf16, index copy back, bf16 add): the same MFMA products on the same values in the same order ->
    bit-identical.  Unaligned text length, head counts that do not divide by six, grids whose window (h * w positions) truncates."""
    from easyanimate_amd import _lib
    from easyanimate_amd.processor import EasyAnimateSWAttnProcessor2_0
    ops = _ops()
    F_, Hh, Ww = grid
    N, B = F_ * Hh * Ww, 2
    S = T + N
    s_pad = ops.round_up(S, 256)
    g = torch.Generator(device="cpu").manual_seed(11)
    q = torch.zeros(B, H, s_pad, 64, dtype=torch.bfloat16, device=DEV)
    k = torch.zeros_like(q)
    vt = torch.zeros(B, H, 64, s_pad, dtype=torch.bfloat16, device=DEV)
    q[:, :, :S] = (torch.randn(B, H, S, 64, generator=g) * ops.FOLDED_Q_SCALE).to(torch.bfloat16).to(DEV)
    k[:, :, :S] = torch.randn(B, H, S, 64, generator=g).to(torch.bfloat16).to(DEV)
    vt[:, :, :, :S] = torch.randn(B, H, 64, S, generator=g).to(torch.bfloat16).to(DEV)
    proc = EasyAnimateSWAttnProcessor2_0(cross_attention_size=cross_size)
    outs = {}
    for copies in (True, False):
        proc.index_copies = copies
        _lib.reset_counters()
        outs[copies] = proc._swa(q, k, vt, B, H, 0, H, T, N, DEV, grid)
        torch.cuda.synchronize()
        c = _lib.counters()
        assert (c.get("attention_window_mapped", 0), c.get("permute_cols", 0), c.get("attention_window", 0)) == ((0, 0, 1) if copies else (1, 1, 0)), c
    assert torch.isfinite(outs[False].float()).all()
    assert torch.equal(outs[False], outs[True])
    again = proc._swa(q, k, vt, B, H, 0, H, T, N, DEV, grid)
    assert torch.equal(again, outs[False])


# ---- round 4: K-blocked operands for the feed-forward pair -------------------------------------------------------------------------
@pytest.mark.parametrize("B,M,dim,inner", [(1, 140000, 256, 1024), (2, 70000, 512, 2048), (1, 66000 + 77, 1024, 1536)])
def test_gemm_kblocked_ffn_pair_equals_row_major(B, M, dim, inner):
    """ea_gemm_bf16_kblocked: the first FFN GEMM writing its GELU output K-blocked ([B, inner / 64, M, 64]) and the second one
    reading it K-blocked together with a K-blocked weight, against the same two GEMMs on row-major operands: the same products
    in the same order -> bit-identical; the blocked intermediate equals the permuted row-major one; ragged M (a last tile of 77
    rows), batch 2 with its own block stride."""
    from easyanimate_amd import _lib
    ops = _ops()
    g = torch.Generator(device="cpu").manual_seed(19)
    x = _bf(torch.randn(B, M, dim, generator=g)).to(DEV)
    w1 = _bf(torch.randn(inner, dim, generator=g) / math.sqrt(dim)).to(DEV)
    w2 = _bf(torch.randn(dim, inner, generator=g) / math.sqrt(inner)).to(DEV)
    b1, b2 = torch.randn(inner, generator=g).to(DEV), torch.randn(dim, generator=g).to(DEV)
    res = _bf(torch.randn(B, M, dim, generator=g)).to(DEV)
    gate = torch.randn(B, dim, generator=g).to(DEV)
    _lib.set_option("gemm_tile", 256)
    try:
        h_ref = ops.gemm(x, w1, b1, ops.EPI_BIAS_GELU_TANH)
        y_ref = ops.gemm(h_ref, w2, b2, ops.EPI_BIAS_GATE_RES, res=res, gate=gate)
        y_ref0 = ops.gemm(h_ref, w2, b2, ops.EPI_BIAS)
    finally:
        _lib.set_option("gemm_tile", 0)
    assert ops.kblocked_ok(B, M, inner, dim) and ops.kblocked_ok(B, M, dim, inner)
    _lib.reset_counters()
    h = ops.gemm_kblocked(x, w1, b1, ops.EPI_BIAS_GELU_TANH, ops.LAYOUT_C)
    w2b = ops.to_kblocked(w2)
    y = ops.gemm_kblocked(h, w2b, b2, ops.EPI_BIAS_GATE_RES, ops.LAYOUT_A | ops.LAYOUT_W, res=res, gate=gate)
    y0 = ops.gemm_kblocked(h, w2b, b2, ops.EPI_BIAS, ops.LAYOUT_A | ops.LAYOUT_W)
    y1 = ops.gemm_kblocked(h, w2, b2, ops.EPI_BIAS, ops.LAYOUT_A)              # blocked A with a row-major weight
    torch.cuda.synchronize()
    assert _lib.counters() == {"gemm_256_w4a": 4}, _lib.counters()
    assert h.shape == (B, inner // 64, M, 64)
    assert torch.equal(h.permute(0, 2, 1, 3).reshape(B, M, inner), h_ref)
    assert torch.equal(y, y_ref) and torch.equal(y0, y_ref0) and torch.equal(y1, y_ref0)


def test_feed_forward_kblocked_path_equals_row_major():
    """attention.FeedForward: the K-blocked pair (default) against EA_KBLOCKED_FFN=0 on the same module -- bit-identical, and the
    blocked path is the one taken at a config-2-sized stream."""
    from easyanimate_amd import _lib, attention
    g = torch.Generator(device="cpu").manual_seed(4)
    ff = attention.FeedForward(512, activation_fn="gelu-approximate", final_dropout=True).to(torch.bfloat16).to(DEV).eval()
    x = _bf(torch.randn(2, 70000, 512, generator=g)).to(DEV)
    res = _bf(torch.randn(2, 70000, 512, generator=g)).to(DEV)
    gate = torch.randn(2, 1, 512, generator=g).to(DEV)
    outs = {}
    for blocked in (True, False):
        attention.KBLOCKED_FFN = blocked
        try:
            with torch.no_grad():
                outs[blocked] = (ff(x, residual=res, gate=gate), ff(x))
        finally:
            attention.KBLOCKED_FFN = True
    assert torch.equal(outs[True][0], outs[False][0]) and torch.equal(outs[True][1], outs[False][1])


def test_feed_forward_kblocked_copy_observes_data_writes():
    """fc2 of the K-blocked pair is read through a derived copy; a `weight.data += delta` LoRA merge
    (utils/lora_utils.py:369-433) between two forwards changes neither data_ptr nor _version.  Outside weights_frozen() the copy is
    re-derived on every call, so forward -> .data merge -> forward equals a module that was loaded with the merged weight; inside
    a frozen region (the pipelines' loops) the copy is kept, and the pipelines drop it once per call."""
    from easyanimate_amd import _lib, _params, attention
    g = torch.Generator(device="cpu").manual_seed(5)
    ff = attention.FeedForward(512, activation_fn="gelu-approximate", final_dropout=True).to(torch.bfloat16).to(DEV).eval()
    x = _bf(torch.randn(2, 70000, 512, generator=g)).to(DEV)
    delta = _bf(0.05 * torch.randn(512, 4, generator=g) @ torch.randn(4, 2048, generator=g)).to(DEV)
    with torch.no_grad():
        _lib.reset_counters()
        y0 = ff(x)
        assert _lib.counters() == {"gemm_256_w4a": 2}, _lib.counters()        # the K-blocked pair is the path under test
        w_orig = ff.net[2].weight.data.clone()
        ff.net[2].weight.data += delta
        y1 = ff(x)
        ref = attention.FeedForward(512, activation_fn="gelu-approximate", final_dropout=True).to(torch.bfloat16).to(DEV).eval()
        ref.load_state_dict(ff.state_dict())
        y_ref = ref(x)
        assert torch.equal(y1, y_ref) and not torch.equal(y1, y0)
        with _params.weights_frozen():
            ff.net[2].weight.data.copy_(w_orig)     # a write the frozen region promises not to do: the copy is NOT refreshed ...
            assert torch.equal(ff(x), y1)
        _params.drop_tag("kblock")                  # ... until the next pipeline call drops it (pipeline.denoise / __call__)
        with _params.weights_frozen():
            assert torch.equal(ff(x), y0)


# ---- round 5: the four-wave 256 x 256 kernel with the hand-placed main loop (gemm256_w4a_kernel) ----------------------------------------
W4A_SHAPES = [
    # B, M, N, K     nk = 1 / 2 / 3 (prologue-only, no in-loop request, first in-loop request), tails in M and N, batches, long K
    (1, 256, 256, 64), (1, 256, 256, 128), (1, 512, 512, 192), (2, 700, 768, 3072), (1, 16500, 320, 320), (2, 1000, 3072, 1024),
    (1, 2048, 12288, 256), (1, 1283, 3072, 12288),
]


@pytest.mark.parametrize("B,M,N,K", W4A_SHAPES)
@pytest.mark.parametrize("epi", [0, 1, 2])
def test_gemm_w4a_bit_identical_to_the_eight_wave_kernel(B, M, N, K, epi):
    """ea_set_option("gemm_w4a", 1): same LDS image, same fragments, the same MFMAs accumulating the same k32 steps in the same
    order per accumulator -> bit-identical to gemm256_mi16_kernel (which is pinned to fp64 by test_gemm_tile256); strided A
    (row stride K + 8, batch gap), ragged last tiles, the gated-residual epilogue in place."""
    from easyanimate_amd import _lib
    ops = _ops()
    g = torch.Generator(device="cpu").manual_seed(41)
    Abuf = _bf(torch.randn(B, M + 3, K + 8, generator=g)).to(DEV)
    A = Abuf[:, 1:M + 1, :K]
    W = _bf(torch.randn(N, K, generator=g) / math.sqrt(K)).to(DEV)
    bias = torch.randn(N, generator=g).to(DEV)
    res = _bf(torch.randn(B, M, N, generator=g)).to(DEV)
    gate = torch.randn(B, N, generator=g).to(DEV)
    outs = []
    _lib.set_option("gemm_tile", 256)
    try:
        for w4a in (0, 1, 1):
            _lib.set_option("gemm_w4a", w4a)
            _lib.reset_counters()
            if epi == 2:
                out = res.clone()
                y = ops.gemm(A, W, bias, epi, out=out, res=out, gate=gate)
            else:
                y = ops.gemm(A, W, bias, epi)
            torch.cuda.synchronize()
            assert _lib.counters() == ({"gemm_256_w4a": 1} if w4a else {"gemm_256_mi16": 1}), _lib.counters()
            outs.append(y.clone())
    finally:
        _lib.set_option("gemm_tile", 0)
        _lib.set_option("gemm_w4a", 3)
    assert torch.isfinite(outs[1].float()).all()
    if not torch.equal(outs[1], outs[0]):
        d = (outs[1].float() - outs[0].float()).abs()
        bad = (d > 0).nonzero()
        raise AssertionError(f"w4a differs from mi16: {bad.shape[0]} of {d.numel()} elements, max {d.max().item():.3e}, first {bad[:5].tolist()}, "
                             f"last {bad[-3:].tolist()}; rows with a difference {bad[:, 1].unique()[:20].tolist()}; cols {bad[:, 2].unique()[:20].tolist()}")
    assert torch.equal(outs[2], outs[1])          # repeated launches agree (race screen)


@pytest.mark.parametrize("epi", [0, 2])
def test_gemm_w4a_without_bias(epi):
    """The four-wave kernel fetches its bias / gate vectors inside the main asm through buffer descriptors: a missing bias is a
    descriptor with num_records = 0 (zeros, no memory touched) -- against the eight-wave kernel and an fp32 product."""
    from easyanimate_amd import _lib
    ops = _ops()
    g = torch.Generator(device="cpu").manual_seed(53)
    B, M, N, K = 2, 700, 768, 256
    A = _bf(torch.randn(B, M, K, generator=g)).to(DEV)
    W = _bf(torch.randn(N, K, generator=g) / math.sqrt(K)).to(DEV)
    res = _bf(torch.randn(B, M, N, generator=g)).to(DEV)
    gate = torch.randn(B, N, generator=g).to(DEV)
    outs = []
    _lib.set_option("gemm_tile", 256)
    try:
        for w4a in (0, 3):
            _lib.set_option("gemm_w4a", w4a)
            _lib.reset_counters()
            y = ops.gemm(A, W, None, epi, res=res if epi == 2 else None, gate=gate if epi == 2 else None)
            torch.cuda.synchronize()
            assert _lib.counters() == ({"gemm_256_w4a": 1} if w4a else {"gemm_256_mi16": 1}), _lib.counters()
            outs.append(y.clone())
    finally:
        _lib.set_option("gemm_tile", 0)
        _lib.set_option("gemm_w4a", 3)
    assert torch.equal(outs[0], outs[1])
    ref = A.float() @ W.float().t()
    if epi == 2:
        ref = res.float() + gate[:, None, :] * ref
    assert (outs[1].float() - ref).abs().max().item() < 2 ** -6 * ref.abs().max().item()


@pytest.mark.parametrize("B,M,dim,inner", [(2, 70000, 512, 2048), (1, 66000 + 77, 1024, 1536)])
def test_gemm_w4a_kblocked_pair_bit_identical(B, M, dim, inner):
    """The K-blocked feed-forward pair (K-blocked C out of the first GEMM, K-blocked A / W into the second) through the four-wave
    kernel: the buffer-addressed requests with their K step as the scalar offset."""
    from easyanimate_amd import _lib
    ops = _ops()
    g = torch.Generator(device="cpu").manual_seed(43)
    x = _bf(torch.randn(B, M, dim, generator=g)).to(DEV)
    w1 = _bf(torch.randn(inner, dim, generator=g) / math.sqrt(dim)).to(DEV)
    w2 = _bf(torch.randn(dim, inner, generator=g) / math.sqrt(inner)).to(DEV)
    b1, b2 = torch.randn(inner, generator=g).to(DEV), torch.randn(dim, generator=g).to(DEV)
    res = _bf(torch.randn(B, M, dim, generator=g)).to(DEV)
    gate = torch.randn(B, dim, generator=g).to(DEV)
    w2b = ops.to_kblocked(w2)
    outs = {}
    try:
        for w4a in (0, 1):
            _lib.set_option("gemm_w4a", w4a)
            _lib.reset_counters()
            h = ops.gemm_kblocked(x, w1, b1, ops.EPI_BIAS_GELU_TANH, ops.LAYOUT_C)
            y = ops.gemm_kblocked(h, w2b, b2, ops.EPI_BIAS_GATE_RES, ops.LAYOUT_A | ops.LAYOUT_W, res=res, gate=gate)
            torch.cuda.synchronize()
            assert _lib.counters() == ({"gemm_256_w4a": 2} if w4a else {"gemm_256_mi16": 2}), _lib.counters()
            outs[w4a] = (h, y)
    finally:
        _lib.set_option("gemm_w4a", 3)
    assert torch.equal(outs[1][0], outs[0][0]) and torch.equal(outs[1][1], outs[0][1])


@pytest.mark.parametrize("B,H,M,K,seq_off,use_rope", [(2, 4, 512, 128, 8, True), (1, 4, 256, 64, 0, False), (2, 48, 768, 3072, 256, True),
                                                      (1, 48, 2016, 3072, 256, True), (2, 8, 1283, 256, 64, True), (1, 4, 13104, 256, 64, True)])
def test_qkv_fused_w4a_bit_identical_to_the_eight_wave_kernel(B, H, M, K, seq_off, use_rope):
    """gemm256_qkv_w4a_kernel (four waves, hand-placed main loop, two heads per wave tile, V tiles on the operand-swapped loop)
    against gemm256_qkv_kernel: the same MFMAs on the same fragments and the same epilogue code per head -> V^T bit-identical,
    q / k to one bf16 ulp in a few elements (FMA contraction differs between the two compiled epilogues), including ragged last tiles (M = 1283 / 13104 / 2016), K | V written into an exchange slot of another geometry
    (kv_off / kv_rows) and the split launch (K | V thirds, then the Q third); nothing outside the addressed rows is written."""
    from easyanimate_amd import _lib
    ops = _ops()
    g = torch.Generator(device="cpu").manual_seed(29)
    d = H * 64
    x = _bf(torch.randn(B, M, K, generator=g)).to(DEV)
    ws = [_bf(torch.randn(d, K, generator=g) / K ** 0.5).to(DEV) for _ in range(3)]
    bs = [(0.3 * torch.randn(d, generator=g)).to(DEV) for _ in range(3)]
    nq_w, nk_w = [(1 + 0.2 * torch.randn(64, generator=g)).to(DEV) for _ in range(2)]
    nq_b, nk_b = [(0.2 * torch.randn(64, generator=g)).to(DEV) for _ in range(2)]
    ang = torch.rand(M, 32, generator=g) * 6.28
    cos = ang.cos().repeat_interleave(2, 1).contiguous().to(DEV) if use_rope else None
    sin = ang.sin().repeat_interleave(2, 1).contiguous().to(DEV) if use_rope else None
    kv_off = seq_off + 64
    s_pad, kv_rows = ops.round_up(seq_off + M, 256), ops.round_up(kv_off + M + 64, 256)
    full = lambda *shape: torch.full(shape, 7.0, dtype=torch.bfloat16, device=DEV)
    outs = {}
    try:
        for w4a in (0, 1):
            _lib.set_option("gemm_w4a", 3 * w4a)
            res = []
            for split in (False, True):
                q, k, vt = full(B, H, s_pad, 64), full(B, H, kv_rows, 64), full(B, H, 64, kv_rows)
                _lib.reset_counters()
                for parts in ((ops.QKV_KV, ops.QKV_Q) if split else (None,)):
                    kw = {} if parts is None else {"parts": parts}
                    ops.qkv_gemm_norm_rope(x, ws[0], ws[1], ws[2], bs[0], bs[1], bs[2], q, k, vt, nq_w, nq_b, nk_w, nk_b, cos, sin, seq_off, 1e-6,
                                           q_scale=ops.FOLDED_Q_SCALE, kv_off=kv_off, **kw)
                torch.cuda.synchronize()
                c = _lib.counters()
                assert c.get("gemm_qkv_fused_w4a", 0) == ((2 if split else 1) if w4a else 0), c
                res.append((q, k, vt))
            assert all(torch.equal(a, b) for a, b in zip(res[0], res[1]))      # split launch == single launch
            outs[w4a] = res[0]
    finally:
        _lib.set_option("gemm_w4a", 3)
    # V^T (bias add + one rounding) is bit-identical.  q / k go through LayerNorm + RoPE arithmetic that the compiler contracts into
    # FMAs per kernel: the two kernels may round a handful of values to neighbouring bf16 numbers (first GPU run: 11 of 6.3 M
    # elements, one ulp) -- the same tolerance the eight-wave kernel has against the unfused path (test_qkv_fused_matches_unfused)
    assert torch.equal(outs[1][2], outs[0][2])
    for name, a, b in zip("q k".split(), outs[1], outs[0]):
        dd = (a.float() - b.float()).abs()
        n_bad = int((dd > 0).sum().item())
        ulp = 2.0 ** -6 * torch.maximum(a.float().abs(), b.float().abs())      # one bf16 ulp at a binade edge
        print(f"[parity] fused QKV four-wave vs eight-wave, {name}: {n_bad} of {dd.numel()} elements differ, max |d| {dd.max().item():.3e}")
        assert n_bad <= max(4, dd.numel() // 20000) and bool((dd <= ulp + 1e-30).all()), (name, n_bad, dd.max().item())
    q, k, vt = outs[1]
    assert torch.isfinite(q[:, :, seq_off:seq_off + M].float()).all()
    assert (q[:, :, :seq_off] == 7).all() and (q[:, :, seq_off + M:] == 7).all()
    assert (k[:, :, :kv_off] == 7).all() and (k[:, :, kv_off + M:] == 7).all() and (vt[:, :, :, :kv_off] == 7).all() and (vt[:, :, :, kv_off + M:] == 7).all()


# ---- round 5: head groups of the pipelined sequence-parallel exchange ------------------------------------------------------------------------
@pytest.mark.parametrize("B,H,G,T,nl,P", [(1, 8, 2, 64, 192, 3), (2, 12, 3, 128, 128, 2), (1, 48, 2, 256, 320, 4)])
def test_attention_head_window_equals_full_launch(B, H, G, T, nl, P):
    """ea_attention_fwd_range_heads_bf16 / ea_attention_fwd_segments_heads_bf16: G launches over head windows, each with the K / V^T of
    its own head group only ([.., H / G, ..] buffers), against one launch over all heads on the un-grouped buffers: the same
    tiles in the same order per head -> out and the carried state bit-identical (own-slot pass with stored state, remote pass
    over P - 1 segments resuming it)."""
    ops = _ops()
    g = torch.Generator(device="cpu").manual_seed(5 + H + P)
    rows = ops.round_up(T + nl, 256)
    S_q = T + nl
    rank = 1
    q = torch.zeros(B, H, rows, 64, dtype=torch.bfloat16, device=DEV)
    q[:, :, :S_q] = _bf(torch.randn(B, H, S_q, 64, generator=g) * 0.3).to(DEV)
    buf = torch.zeros(P, 2, B, H, rows * 64, dtype=torch.bfloat16, device=DEV)
    for r in range(P):
        buf[r, 0].view(B, H, rows, 64)[:, :, :S_q] = _bf(torch.randn(B, H, S_q, 64, generator=g)).to(DEV)
        buf[r, 1].view(B, H, 64, rows)[:, :, :, :S_q] = _bf(torch.randn(B, H, 64, S_q, generator=g)).to(DEV)
    Hg = H // G
    bufg = torch.zeros(G, P, 2, B, Hg, rows * 64, dtype=torch.bfloat16, device=DEV)
    for gi in range(G):
        bufg[gi] = buf.view(P, 2, B, G, Hg, rows * 64)[:, :, :, gi]
    remote_valid = (P - 1) * nl
    st_a, st_b = ops.attention_state(B, H, 0, S_q, DEV), ops.attention_state(B, H, 0, S_q, DEV)
    o_a, o_b = (torch.full((B, S_q, H * 64), 7.0, dtype=torch.bfloat16, device=DEV) for _ in range(2))
    k_own, vt_own = buf[rank, 0].view(B, H, rows, 64), buf[rank, 1].view(B, H, 64, rows)
    ops.attention_range(q, k_own, vt_own, ops.FOLDED_ATTN_SCALE, 0, S_q, 0, S_q, state=st_a, store_state=True)
    ops.attention_segments(q, buf, P, rank, rows, remote_valid, 0, S_q, state=st_a, load_state=True, out=o_a, first_row=T, used_rows=nl)
    for gi in range(G):
        kg, vg = bufg[gi, rank, 0].view(B, Hg, rows, 64), bufg[gi, rank, 1].view(B, Hg, 64, rows)
        ops.attention_range(q, kg, vg, ops.FOLDED_ATTN_SCALE, 0, S_q, 0, S_q, state=st_b, store_state=True, head0=gi * Hg)
    torch.cuda.synchronize()
    assert torch.equal(st_a, st_b)
    for gi in range(G):
        ops.attention_segments(q, bufg[gi], P, rank, rows, remote_valid, 0, S_q, state=st_b, load_state=True, out=o_b, first_row=T, used_rows=nl,
                               head0=gi * Hg, group_heads=Hg)
    torch.cuda.synchronize()
    assert torch.isfinite(o_b.float()).all() and torch.equal(o_a, o_b)


@pytest.mark.parametrize("B,H,G,M,K,seq_off", [(2, 8, 2, 512, 128, 64), (1, 48, 2, 1283, 256, 256), (2, 12, 3, 768, 64, 0)])
def test_qkv_grouped_destination(B, H, G, M, K, seq_off):
    """ea_qkv_gemm_norm_rope_grouped_bf16: K / V^T written per head group into [G, .., B, H / G, rows, 64] buffers (the own slots of
    the group-wise exchange buffers) against the un-grouped launch: bit-identical values at the grouped addresses, q unchanged,
    nothing else written; both launch forms (one launch; K | V then Q)."""
    ops = _ops()
    g = torch.Generator(device="cpu").manual_seed(61 + H)
    d = H * 64
    x = _bf(torch.randn(B, M, K, generator=g)).to(DEV)
    ws = [_bf(torch.randn(d, K, generator=g) / K ** 0.5).to(DEV) for _ in range(3)]
    bs = [(0.3 * torch.randn(d, generator=g)).to(DEV) for _ in range(3)]
    n4 = [(1 + 0.2 * torch.randn(64, generator=g)).to(DEV), (0.2 * torch.randn(64, generator=g)).to(DEV),
          (1 + 0.2 * torch.randn(64, generator=g)).to(DEV), (0.2 * torch.randn(64, generator=g)).to(DEV)]
    ang = torch.rand(M, 32, generator=g) * 6.28
    cos, sin = ang.cos().repeat_interleave(2, 1).contiguous().to(DEV), ang.sin().repeat_interleave(2, 1).contiguous().to(DEV)
    kv_off = seq_off + 64
    s_pad, kv_rows = ops.round_up(seq_off + M, 256), ops.round_up(kv_off + M + 64, 256)
    full = lambda *shape: torch.full(shape, 7.0, dtype=torch.bfloat16, device=DEV)
    q0, k0, vt0 = full(B, H, s_pad, 64), full(B, H, kv_rows, 64), full(B, H, 64, kv_rows)
    ops.qkv_gemm_norm_rope(x, ws[0], ws[1], ws[2], bs[0], bs[1], bs[2], q0, k0, vt0, n4[0], n4[1], n4[2], n4[3], cos, sin, seq_off, 1e-6,
                           q_scale=ops.FOLDED_Q_SCALE, kv_off=kv_off)
    Hg, P, rank = H // G, 3, 1
    for split in (False, True):
        q1 = full(B, H, s_pad, 64)
        kvg = full(G, P, 2, B, Hg, kv_rows * 64)                 # group-wise exchange buffers; the launch writes into slot `rank` of each
        k_own, vt_own = kvg[0, rank, 0].view(B, Hg, kv_rows, 64), kvg[0, rank, 1].view(B, Hg, 64, kv_rows)
        for parts in ((ops.QKV_KV, ops.QKV_Q) if split else (ops.QKV_ALL,)):
            ops.qkv_gemm_norm_rope(x, ws[0], ws[1], ws[2], bs[0], bs[1], bs[2], q1, k_own, vt_own, n4[0], n4[1], n4[2], n4[3], cos, sin, seq_off, 1e-6,
                                   q_scale=ops.FOLDED_Q_SCALE, kv_off=kv_off, parts=parts, kv_group_stride=kvg.stride(0))
        torch.cuda.synchronize()
        assert torch.equal(q1, q0)
        for gi in range(G):
            kg, vg = kvg[gi, rank, 0].view(B, Hg, kv_rows, 64), kvg[gi, rank, 1].view(B, Hg, 64, kv_rows)
            assert torch.equal(kg, k0[:, gi * Hg:(gi + 1) * Hg]) and torch.equal(vg, vt0[:, gi * Hg:(gi + 1) * Hg])
        others = [r for r in range(P) if r != rank]
        assert (kvg[:, others] == 7).all()
