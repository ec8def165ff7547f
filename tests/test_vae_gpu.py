"""GPU parity of the VAE kernels (vs plain PyTorch on the same bf16-rounded inputs) and of the product
AutoencoderKLMagvit (vs golden vectors from the reference's chunked mode and the oracle restatement)."""
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _eight_wave_conv_dispatch():
    """The kernel-level tests of this file pin the dispatch and the arithmetic of the EIGHT-wave convolution kernels (exact launch
    counters, forced tile options): they run with conv_w4a = 0.  The four-wave hand-placed kernels -- the default since round 5
    for the 512 x 128 and 256 x 256 row-slab tiles -- are covered by test_conv_w4a_bit_identical_to_the_eight_wave_kernels below
    (which sets the option itself) and by every model-level golden (tests/test_parity_r2_gpu.py, test_config4_gpu.py)."""
    from easyanimate_amd import _lib
    prev = _lib.get_option("conv_w4a")
    _lib.set_option("conv_w4a", 0)
    yield
    _lib.set_option("conv_w4a", prev)
DEV = "cuda"
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _rep(name, got, ref):
    got, ref = got.double().cpu(), ref.double().cpu()
    err = (got - ref).abs().max().item()
    rel = ((got - ref).norm() / (ref.norm() + 1e-30)).item()
    print(f"[parity] {name}: max_abs={err:.3e} rel_l2={rel:.3e} ref_absmax={ref.abs().max().item():.3e}")
    return err, rel


def _bf(x):
    return x.to(torch.bfloat16)


CONV_CASES = [
    # T, H, W, Cin, Cout, k, st, ss, pad, ups, tdup, res
    (3, 10, 12, 64, 64, 3, 1, 1, 1, False, False, False),
    (4, 9, 7, 128, 72, 3, 1, 1, 1, False, False, True),    # ragged M / N tails + residual
    (5, 16, 12, 64, 128, 3, 1, 2, 0, False, False, False),  # SpatialDownsampler3D
    (5, 16, 12, 64, 64, 3, 2, 2, 0, False, False, False),   # SpatialTemporalDownsampler3D
    (9, 8, 8, 128, 64, 3, 2, 2, 0, False, False, False),
    (3, 6, 5, 64, 64, 3, 1, 1, 1, True, False, False),     # SpatialUpsampler3D (nearest x2 folded)
    (3, 6, 5, 64, 64, 3, 1, 1, 1, True, True, False),      # SpatialTemporalUpsampler3D (temporal dup)
    (1, 6, 6, 64, 64, 3, 1, 1, 1, True, True, False),      # single frame: no temporal dup
    (4, 5, 6, 128, 256, 1, 1, 1, 0, False, False, False),  # 1x1x1 shortcut
    (2, 40, 40, 256, 256, 3, 1, 1, 1, False, False, True),
]


@pytest.mark.parametrize("T,H,W,Ci,Co,k,st,ss,pad,ups,tdup,res", CONV_CASES)
def test_conv3d_cl(T, H, W, Ci, Co, k, st, ss, pad, ups, tdup, res):
    from easyanimate_amd import ops
    from easyanimate_amd.vae_modules import _pack_conv_weight
    g = torch.Generator().manual_seed(3)
    x = _bf(torch.randn(1, Ci, T, H, W, generator=g))
    w = _bf(torch.randn(Co, Ci, k, k, k, generator=g) / (Ci * k ** 3) ** 0.5)
    b = torch.randn(Co, generator=g)
    # reference (fp64): causal replicate pad in time, nearest x2, asymmetric pad for the strided convs
    xr = x.double()
    if ups:
        xr = F.interpolate(xr, scale_factor=(1, 2, 2), mode="nearest")
    if k == 3 and pad == 0:
        xr = F.pad(xr, (0, 1, 0, 1))
    xr = F.pad(xr, (0, 0, 0, 0, k - 1, 0), mode="replicate")
    ref = F.conv3d(xr, w.double(), b.double(), stride=(st, ss, ss), padding=(0, pad, pad))
    r = None
    if res:
        r = _bf(torch.randn(ref.shape, generator=g))
        ref = ref + r.double()
    if tdup and ref.shape[2] > 1:
        ref = torch.cat([ref[:, :, :1], F.interpolate(ref[:, :, 1:], scale_factor=(2, 1, 1), mode="nearest")], 2)
    xcl = x[0].permute(1, 2, 3, 0).contiguous().to(DEV)
    rcl = None if r is None else r[0].permute(1, 2, 3, 0).contiguous().to(DEV)
    y = ops.conv3d_cl(xcl, _pack_conv_weight(w).to(DEV), b.to(DEV), k, st, ss, pad, ups=ups, tdup=tdup, res=rcl)
    got = y.permute(3, 0, 1, 2)[None]
    assert got.shape == ref.shape, (got.shape, ref.shape)
    err, rel = _rep(f"conv3d_cl T{T} {H}x{W} {Ci}->{Co} k{k} s{st}{ss} ups{int(ups)} tdup{int(tdup)}", got, ref)
    assert rel < 4e-3


@pytest.mark.parametrize("T,H,W,Ci,Co", [(3, 10, 12, 64, 3), (1, 7, 9, 128, 3), (5, 33, 24, 128, 3), (2, 8, 8, 64, 1), (4, 6, 5, 192, 4)])
def test_conv3d_narrow_n(T, H, W, Ci, Co):
    """Decoder conv_out (128 -> 3, omnigen_enc_dec.py:611): one GEMM over the input voxels + the 27-tap gather
    (ea_conv3d_tap_gather_f32) against fp64 F.conv3d with causal replicate padding -- through the module's own dispatch."""
    from easyanimate_amd import _lib
    from easyanimate_amd.vae_modules import CausalConv3d
    g = torch.Generator().manual_seed(11)
    conv = CausalConv3d(Ci, Co, kernel_size=3)
    with torch.no_grad():
        conv.weight.copy_(_bf(torch.randn(Co, Ci, 3, 3, 3, generator=g) / (Ci * 27) ** 0.5).float())
        conv.bias.copy_(torch.randn(Co, generator=g))
    x = _bf(torch.randn(1, Ci, T, H, W, generator=g))
    ref = F.conv3d(F.pad(x.double(), (0, 0, 0, 0, 2, 0), mode="replicate"), conv.weight.double(), conv.bias.double(), padding=(0, 1, 1))
    conv = conv.to(DEV)
    _lib.reset_counters()
    y = conv(x[0].permute(1, 2, 3, 0).contiguous().to(DEV))
    # voxel counts that are not a multiple of 8 (the GEMM's column granularity) keep the padded tile-per-tap kernel
    assert _lib.counters().get("conv_narrow_gemm_tap_gather", 0) == (1 if (T * H * W) % 8 == 0 else 0)
    assert y.shape == (T, H, W, 8) and y[..., Co:].abs().max().item() == 0
    err, rel = _rep(f"conv3d narrow-N T{T} {H}x{W} {Ci}->{Co}", y[..., :Co].permute(3, 0, 1, 2)[None], ref)
    assert rel < 4e-3


@pytest.mark.parametrize("Ci,Co,ups,res,tdup", [(128, 128, False, True, False), (64, 256, False, False, False), (128, 128, True, False, False),
                                                 (64, 256, True, False, True)])
def test_groupnorm_stats_fused_in_conv_epilogue(Ci, Co, ups, res, tdup):
    """The row-slab convolution leaves the per-frame (sum, sumsq) partials of its output behind (ea_conv3d_cl_stats_bf16);
    GroupNorm + SiLU from those partials (finalize only) must equal GroupNorm + SiLU with its own statistics pass."""
    from easyanimate_amd import _lib, ops
    from easyanimate_amd.vae_modules import _pack_conv_weight
    g = torch.Generator().manual_seed(23)
    T, H, W = 3, (128 if ups else 256), (128 if ups else 256)
    x = _bf(torch.randn(T, H, W, Ci, generator=g)).to(DEV)
    w = _pack_conv_weight(_bf(torch.randn(Co, Ci, 3, 3, 3, generator=g) / (Ci * 27) ** 0.5)).to(DEV)
    b = torch.randn(Co, generator=g).to(DEV)
    r = _bf(torch.randn(T, 256, 256, Co, generator=g)).to(DEV) if res else None
    gamma, beta = (1 + 0.3 * torch.randn(Co, generator=g)).to(DEV), (0.3 * torch.randn(Co, generator=g)).to(DEV)
    _lib.reset_counters()
    y = ops.conv3d_cl(x, w, b, 3, ups=ups, res=r, tdup=tdup)
    assert sum(v for k, v in _lib.counters().items() if k.startswith("conv_row16")) == 1
    assert hasattr(y, "gn_partial"), "the row-slab kernel served the call but left no partial sums"
    assert y.shape[0] == (2 * T - 1 if tdup else T)
    y_plain = ops.conv3d_cl(x, w, b, 3, ups=ups, res=r, tdup=tdup, want_stats=False)
    assert torch.equal(y, y_plain) and not hasattr(y_plain, "gn_partial")
    a = ops.groupnorm_silu(y, gamma, beta, 32, 1e-6)
    ops.FUSED_GN_STATS = False
    try:
        c = ops.groupnorm_silu(y, gamma, beta, 32, 1e-6)
    finally:
        ops.FUSED_GN_STATS = True
    ref = F.silu(F.group_norm(y.double().permute(0, 3, 1, 2), 32, gamma.double(), beta.double(), 1e-6)).permute(0, 2, 3, 1)
    d = (a.float() - c.float()).abs()
    frac = (d > 0).float().mean().item()
    print(f"[parity] GroupNorm from conv-epilogue partials vs own statistics pass ({Ci}->{Co}, ups {ups}, res {res}, tdup {tdup}): "
          f"{frac * 100:.4f} % of the outputs differ, max |d| {d.max().item():.3e}")
    assert frac < 1e-3 and d.max().item() <= 2.0 ** -6 * max(1.0, c.float().abs().max().item())
    err, rel = _rep("GroupNorm(fused stats)+SiLU vs fp64", a, ref)
    assert rel < 5e-3


PP_CASES = [
    # the 256 x {128,256} ping-pong kernels, forced: ragged M, taps crossing every border, strides, folded up-sampling,
    # temporal dup, residual, 1x1x1, odd / even K-tile counts (C_in 64 -> 27 tiles, 128 -> 54, 192 -> 81)
    (3, 10, 12, 64, 128, 3, 1, 1, 1, False, False, False),
    (4, 9, 7, 128, 256, 3, 1, 1, 1, False, False, True),
    (5, 16, 12, 64, 128, 3, 1, 2, 0, False, False, False),
    (5, 16, 12, 192, 256, 3, 2, 2, 0, False, False, False),
    (3, 6, 5, 64, 128, 3, 1, 1, 1, True, True, True),
    (1, 6, 6, 128, 128, 3, 1, 1, 1, True, True, False),
    (4, 5, 6, 128, 256, 1, 1, 1, 0, False, False, False),
    (2, 40, 40, 256, 256, 3, 1, 1, 1, False, False, True),
    (3, 33, 31, 128, 512, 3, 1, 1, 1, False, False, False),
]


@pytest.mark.parametrize("tile", [256, 512])
@pytest.mark.parametrize("T,H,W,Ci,Co,k,st,ss,pad,ups,tdup,res", PP_CASES)
def test_conv3d_cl_pingpong(T, H, W, Ci, Co, k, st, ss, pad, ups, tdup, res, tile):
    from easyanimate_amd import _lib
    if tile == 512 and Co != 128:
        pytest.skip("the 512-row tile exists for C_out = 128 only")
    _lib.set_option("conv_tile", tile)
    try:
        test_conv3d_cl(T, H, W, Ci, Co, k, st, ss, pad, ups, tdup, res)
    finally:
        _lib.set_option("conv_tile", 0)


def test_conv3d_cl_pingpong_equals_128_bitwise():
    """Race screen for the hand-placed vmcnt / barrier schedule: identical results on repeated launches and bit-equal to
    the 128^2 kernel (same K order, same fp32 accumulation chain) at a VAE-sized layer."""
    from easyanimate_amd import _lib, ops
    from easyanimate_amd.vae_modules import _pack_conv_weight
    g = torch.Generator().manual_seed(8)
    for (Ci, Co) in ((128, 128), (256, 256)):
        x = _bf(torch.randn(5, 96, 96, Ci, generator=g)).to(DEV)
        w = _pack_conv_weight(_bf(torch.randn(Co, Ci, 3, 3, 3, generator=g) / (Ci * 27) ** 0.5)).to(DEV)
        b = torch.randn(Co, generator=g).to(DEV)
        _lib.set_option("conv_tile", 128)
        y0 = ops.conv3d_cl(x, w, b, 3, 1, 1, 1)
        for tile in ((256, 512) if Co == 128 else (256,)):
            _lib.set_option("conv_tile", tile)
            try:
                y1 = ops.conv3d_cl(x, w, b, 3, 1, 1, 1)
                for _ in range(4):
                    assert torch.equal(ops.conv3d_cl(x, w, b, 3, 1, 1, 1), y1)
            finally:
                _lib.set_option("conv_tile", 0)
            assert torch.equal(y0, y1)


ROW_CASES = [
    # the row-slab kernel (forced): rows 256 / 512 / 768 voxels wide, first / last row and frame 0 (causal replicate),
    # temporal dup, residual, one to four channel blocks, C_out 128 / 256 / 512 (N tiles)
    (3, 4, 256, 64, 128, False, False, False),
    (2, 3, 512, 128, 128, False, False, True),
    (3, 2, 256, 128, 256, False, True, True),
    (1, 5, 768, 64, 256, False, False, False),
    (4, 3, 256, 256, 512, False, False, True),
    (2, 1, 256, 192, 128, False, False, False),
    # nearest x2 up-sampling folded into the slab addressing (input rows 128 / 256 / 384 wide -> output 256 / 512 / 768)
    (3, 3, 128, 64, 128, True, True, False),
    (2, 2, 256, 128, 256, True, False, False),
    (1, 3, 128, 128, 128, True, True, False),
    (2, 1, 384, 64, 256, True, True, False),
]


@pytest.mark.parametrize("mfma,k32", [(16, 3), (16, 0), (32, 0)])
@pytest.mark.parametrize("T,H,W,Ci,Co,ups,tdup,res", ROW_CASES)
def test_conv3d_cl_row_slab(T, H, W, Ci, Co, ups, tdup, res, mfma, k32):
    """k32 = the "conv_m512" switch: 3 sends the layers without folded up-sampling to the one-phase-per-tile kernels over
    32-channel stages (512 voxels x 128 channels, 256 x 256), 0 keeps the four-phase kernels over 64-channel stages."""
    from easyanimate_amd import _lib
    if mfma == 32 and _lib.get_option("build_variants") != 1:
        pytest.skip("the 32x32x16 row-slab kernel is a cross-check generation: build with EA_BUILD_VARIANTS=1")
    k0 = _lib.get_option("conv_m512")
    _lib.set_option("conv_tile", 1024)
    _lib.set_option("conv_mfma", mfma)
    _lib.set_option("conv_m512", k32)
    _lib.reset_counters()
    try:
        test_conv3d_cl(T, H, W, Ci, Co, 3, 1, 1, 1, ups, tdup, res)
    finally:
        _lib.set_option("conv_tile", 0)
        _lib.set_option("conv_mfma", 16)
        _lib.set_option("conv_m512", k0)
    c = _lib.counters()
    if mfma == 16 and k32 == 3 and not ups and not tdup and Co % 256 == 0:
        assert c == {"conv_row16_256_k32": 1}, c
    elif mfma == 16 and k32 == 3 and not ups and not tdup and Co == 128 and W % 512 == 0:
        assert c == {"conv_row16_m512": 1}, c
    elif mfma == 16:
        assert len(c) == 1 and next(iter(c)).startswith("conv_row16_") and "k32" not in next(iter(c)) and "m512" not in next(iter(c)), c


def test_conv3d_cl_row_slab_is_deterministic_and_matches_tilewise():
    """Race screen for the row-slab schedule: repeated launches are bit-identical; against the tile-per-tap kernel (a
    different K order) the results agree to fp32 summation-order noise."""
    from easyanimate_amd import _lib, ops
    from easyanimate_amd.vae_modules import _pack_conv_weight
    g = torch.Generator().manual_seed(9)
    for (Ci, Co) in ((128, 128), (256, 256)):
        x = _bf(torch.randn(3, 40, 512, Ci, generator=g)).to(DEV)
        w = _pack_conv_weight(_bf(torch.randn(Co, Ci, 3, 3, 3, generator=g) / (Ci * 27) ** 0.5)).to(DEV)
        b = torch.randn(Co, generator=g).to(DEV)
        _lib.set_option("conv_tile", 256)
        y0 = ops.conv3d_cl(x, w, b, 3, 1, 1, 1)
        _lib.set_option("conv_tile", 1024)
        try:
            for mfma in ((32, 16) if _lib.get_option("build_variants") == 1 else (16,)):
                _lib.set_option("conv_mfma", mfma)
                y1 = ops.conv3d_cl(x, w, b, 3, 1, 1, 1)
                for _ in range(4):
                    assert torch.equal(ops.conv3d_cl(x, w, b, 3, 1, 1, 1), y1)
                d = (y0.float() - y1.float()).abs()
                # a few last-bit bf16 flips only
                assert bool((d <= 2 ** -7 * y0.float().abs().clamp_min(1.0)).all()) and (d > 0).float().mean().item() < 0.01
        finally:
            _lib.set_option("conv_tile", 0)
            _lib.set_option("conv_mfma", 16)


M512_CASES = [
    # the 512-voxel row-slab kernel (C_out = 128, rows a multiple of 512): one / two tiles per row, first / last row,
    # frame 0 (causal replicate), residual, two to six 32-channel stages
    (2, 3, 512, 128, 128, True),
    (3, 2, 1024, 64, 128, False),
    (1, 5, 512, 64, 128, False),
    (2, 1, 1536, 192, 128, True),
]


@pytest.mark.parametrize("T,H,W,Ci,Co,res", M512_CASES)
def test_conv3d_cl_row_slab_m512(T, H, W, Ci, Co, res):
    from easyanimate_amd import _lib
    _lib.set_option("conv_tile", 1024)
    _lib.reset_counters()
    try:
        test_conv3d_cl(T, H, W, Ci, Co, 3, 1, 1, 1, False, False, res)
    finally:
        _lib.set_option("conv_tile", 0)
    assert _lib.counters().get("conv_row16_m512", 0) == 1


@pytest.mark.parametrize("Ci,Co,W", [(128, 128, 1024), (256, 256, 512), (128, 512, 256)])
def test_conv3d_cl_row_slab_k32_deterministic_stats_and_switch(Ci, Co, W):
    """Race screen (repeated launches bit-identical) for the one-phase-per-tile kernels, their GroupNorm partial sums against
    a statistics pass over the output, and the "conv_m512" switch: 0 serves the same call with the four-phase kernel over
    64-channel stages (another K order: last-bit bf16 flips only)."""
    from easyanimate_amd import _lib, ops
    from easyanimate_amd.vae_modules import _pack_conv_weight
    g = torch.Generator().manual_seed(10)
    x = _bf(torch.randn(3, 24, W, Ci, generator=g)).to(DEV)
    w = _pack_conv_weight(_bf(torch.randn(Co, Ci, 3, 3, 3, generator=g) / (Ci * 27) ** 0.5)).to(DEV)
    b = torch.randn(Co, generator=g).to(DEV)
    gamma, beta = (1 + 0.3 * torch.randn(Co, generator=g)).to(DEV), (0.3 * torch.randn(Co, generator=g)).to(DEV)
    name = "conv_row16_m512" if Co == 128 else "conv_row16_256_k32"
    k0 = _lib.get_option("conv_m512")
    _lib.set_option("conv_tile", 1024)
    _lib.set_option("conv_m512", 3)
    try:
        _lib.reset_counters()
        y1 = ops.conv3d_cl(x, w, b, 3, 1, 1, 1)
        assert _lib.counters() == {name: 1} and hasattr(y1, "gn_partial")
        for _ in range(4):
            assert torch.equal(ops.conv3d_cl(x, w, b, 3, 1, 1, 1), y1)
        a = ops.groupnorm_silu(y1, gamma, beta, 32, 1e-6)
        ops.FUSED_GN_STATS = False
        try:
            c = ops.groupnorm_silu(y1, gamma, beta, 32, 1e-6)
        finally:
            ops.FUSED_GN_STATS = True
        d = (a.float() - c.float()).abs()
        assert (d > 0).float().mean().item() < 1e-3 and d.max().item() <= 2.0 ** -6 * max(1.0, c.float().abs().max().item())
        _lib.set_option("conv_m512", 0)
        _lib.reset_counters()
        y0 = ops.conv3d_cl(x, w, b, 3, 1, 1, 1)
        assert _lib.counters() == {("conv_row16_128" if Co == 128 else "conv_row16_256"): 1}
        d = (y0.float() - y1.float()).abs()
        assert bool((d <= 2 ** -7 * y0.float().abs().clamp_min(1.0)).all()) and (d > 0).float().mean().item() < 0.01
    finally:
        _lib.set_option("conv_m512", k0)
        _lib.set_option("conv_tile", 0)


def test_small_cin_conv_via_im2col():
    from easyanimate_amd.vae_modules import CausalConv3d
    g = torch.Generator().manual_seed(4)
    for ci, co in ((3, 64), (16, 128), (64, 3)):
        conv = CausalConv3d(ci, co, kernel_size=3)
        with torch.no_grad():
            conv.weight.copy_(_bf(torch.randn(conv.weight.shape, generator=g) / (27 * ci) ** 0.5).float())
            conv.bias.copy_(torch.randn(co, generator=g))
        x = _bf(torch.randn(1, ci, 4, 9, 10, generator=g))
        ref = F.conv3d(F.pad(x.double(), (0, 0, 0, 0, 2, 0), mode="replicate"), conv.weight.double(), conv.bias.double(), padding=(0, 1, 1))
        conv = conv.to(DEV)
        y = conv(x[0].permute(1, 2, 3, 0).contiguous().to(DEV))
        got = y[..., :co].permute(3, 0, 1, 2)[None]
        err, rel = _rep(f"conv {ci}->{co}", got, ref)
        assert rel < 4e-3
        if y.shape[-1] != co:
            assert y[..., co:].abs().max().item() == 0


@pytest.mark.parametrize("ci,co,T,H,W,st,ss,pad", [(3, 128, 5, 33, 40, 1, 1, 1), (3, 64, 4, 9, 10, 1, 1, 1), (8, 128, 3, 16, 24, 1, 1, 1),
                                                  (1, 256, 2, 20, 20, 1, 1, 1), (3, 128, 5, 16, 18, 2, 2, 0)])
def test_conv_8_channel_input(ci, co, T, H, W, st, ss, pad):
    """conv3d_cl_kernel<C8> (the encoder's conv_in: RGB handed over as 8 channels, omnigen_enc_dec.py:100-107): eight taps
    per K tile gathered by address, against fp64 F.conv3d with causal replicate padding -- through the module's dispatch
    -- and bit-identical to the im2col + GEMM route it replaces for the same K order?  No: another K layout (tap slots of 8),
    so fp32 summation-order noise only."""
    from easyanimate_amd import _lib, ops
    from easyanimate_amd.vae_modules import CausalConv3d
    g = torch.Generator().manual_seed(41)
    conv = CausalConv3d(ci, co, kernel_size=3, stride=(st, ss, ss), padding=pad)
    with torch.no_grad():
        conv.weight.copy_(_bf(torch.randn(conv.weight.shape, generator=g) / (27 * ci) ** 0.5).float())
        conv.bias.copy_(torch.randn(co, generator=g))
    x = _bf(torch.randn(1, ci, T, H, W, generator=g))
    xr = x.double()
    if pad == 0:
        xr = F.pad(xr, (0, 1, 0, 1))
    xr = F.pad(xr, (0, 0, 0, 0, 2, 0), mode="replicate")
    ref = F.conv3d(xr, conv.weight.double(), conv.bias.double(), stride=(st, ss, ss), padding=(0, pad, pad))
    conv = conv.to(DEV)
    x8 = ops.ncdhw_to_ndhwc(x[0].contiguous().to(DEV), 8)
    assert x8.shape == (T, H, W, 8) and (ci == 8 or x8[..., ci:].abs().max().item() == 0)
    _lib.reset_counters()
    y = conv(x8)
    assert _lib.counters() == {"conv_c8_128x128": 1}
    got = y[..., :co].permute(3, 0, 1, 2)[None]
    assert got.shape == ref.shape
    err, rel = _rep(f"conv 8-channel input {ci}->{co} s{st}{ss}", got, ref)
    assert rel < 4e-3
    for _ in range(3):
        assert torch.equal(conv(x8), y)
    # the im2col + GEMM route on the unpadded layout agrees to summation-order noise
    y2 = conv(x[0].permute(1, 2, 3, 0).contiguous().to(DEV))
    d = (y2.float() - y.float()).abs()
    assert bool((d <= 2 ** -6 * y2.float().abs().clamp_min(1.0)).all())


S2_CASES = [
    # the strided down-sampler on the de-interleaved row-slab kernel: spatial stride 2 (temporal 1 / 2), pad 0 with the zero row
    # / column on the high side (H_in even: the last output row reads it), one and two 512-voxel tiles per output row
    (3, 6, 1024, 64, 1), (4, 5, 1024, 128, 2), (2, 4, 2048, 64, 1), (5, 2, 1024, 192, 2),
]


@pytest.mark.parametrize("T,H,W,Ci,st", S2_CASES)
def test_conv3d_cl_strided_row_slab(T, H, W, Ci, st):
    from easyanimate_amd import _lib
    _lib.reset_counters()
    test_conv3d_cl(T, H, W, Ci, 128, 3, st, 2, 0, False, False, False)
    assert _lib.counters() == {"conv_row16_m512_s2": 1}
    # the switch that turns the one-phase kernels off sends the layer back to the tile-per-tap kernel
    k0 = _lib.get_option("conv_m512")
    _lib.set_option("conv_m512", 0)
    try:
        _lib.reset_counters()
        test_conv3d_cl(T, H, W, Ci, 128, 3, st, 2, 0, False, False, False)
        assert "conv_row16_m512_s2" not in _lib.counters()
    finally:
        _lib.set_option("conv_m512", k0)


def test_conv3d_cl_strided_row_slab_deterministic_and_stats():
    from easyanimate_amd import _lib, ops
    from easyanimate_amd.vae_modules import _pack_conv_weight
    g = torch.Generator().manual_seed(12)
    x = _bf(torch.randn(3, 16, 1024, 128, generator=g)).to(DEV)
    w = _pack_conv_weight(_bf(torch.randn(128, 128, 3, 3, 3, generator=g) / (128 * 27) ** 0.5)).to(DEV)
    b = torch.randn(128, generator=g).to(DEV)
    gamma, beta = (1 + 0.3 * torch.randn(128, generator=g)).to(DEV), (0.3 * torch.randn(128, generator=g)).to(DEV)
    _lib.reset_counters()
    y = ops.conv3d_cl(x, w, b, 3, 1, 2, 0)
    assert _lib.counters() == {"conv_row16_m512_s2": 1} and y.shape == (3, 8, 512, 128) and hasattr(y, "gn_partial")
    for _ in range(4):
        assert torch.equal(ops.conv3d_cl(x, w, b, 3, 1, 2, 0), y)
    a = ops.groupnorm_silu(y, gamma, beta, 32, 1e-6)
    ops.FUSED_GN_STATS = False
    try:
        c = ops.groupnorm_silu(y, gamma, beta, 32, 1e-6)
    finally:
        ops.FUSED_GN_STATS = True
    d = (a.float() - c.float()).abs()
    assert (d > 0).float().mean().item() < 1e-3 and d.max().item() <= 2.0 ** -6 * max(1.0, c.float().abs().max().item())
    k0 = _lib.get_option("conv_m512")
    _lib.set_option("conv_m512", 0)
    try:
        y0 = ops.conv3d_cl(x, w, b, 3, 1, 2, 0)
    finally:
        _lib.set_option("conv_m512", k0)
    d = (y0.float() - y.float()).abs()
    assert bool((d <= 2 ** -7 * y0.float().abs().clamp_min(1.0)).all()) and (d > 0).float().mean().item() < 0.01


def test_conv_8_channel_input_leaves_groupnorm_partials():
    """The encoder's conv_in at a frame size that is a multiple of 128 voxels: the 8-channel kernel's epilogue leaves the
    GroupNorm partial sums of its output; GroupNorm + SiLU from them equals GroupNorm + SiLU with its own statistics pass."""
    from easyanimate_amd import _lib, ops
    from easyanimate_amd.vae_modules import CausalConv3d
    g = torch.Generator().manual_seed(43)
    conv = CausalConv3d(3, 128, kernel_size=3)
    with torch.no_grad():
        conv.weight.copy_(_bf(torch.randn(conv.weight.shape, generator=g) / 9.0).float())
        conv.bias.copy_(torch.randn(128, generator=g))
    conv = conv.to(DEV)
    x8 = ops.ncdhw_to_ndhwc((torch.rand(3, 5, 256, 256, generator=g) * 2 - 1).to(DEV), 8)
    gamma, beta = (1 + 0.3 * torch.randn(128, generator=g)).to(DEV), (0.3 * torch.randn(128, generator=g)).to(DEV)
    _lib.reset_counters()
    y = conv(x8)
    assert _lib.counters() == {"conv_c8_128x128": 1} and hasattr(y, "gn_partial")
    a = ops.groupnorm_silu(y, gamma, beta, 32, 1e-6)
    ops.FUSED_GN_STATS = False
    try:
        c = ops.groupnorm_silu(y, gamma, beta, 32, 1e-6)
    finally:
        ops.FUSED_GN_STATS = True
    d = (a.float() - c.float()).abs()
    assert (d > 0).float().mean().item() < 1e-3 and d.max().item() <= 2.0 ** -6 * max(1.0, c.float().abs().max().item())
    ref = F.silu(F.group_norm(y.double().permute(0, 3, 1, 2), 32, gamma.double(), beta.double(), 1e-6)).permute(0, 2, 3, 1)
    err, rel = _rep("GroupNorm(partials of the 8-channel conv)+SiLU vs fp64", a, ref)
    assert rel < 5e-3


@pytest.mark.parametrize("T,HW,C,G", [(3, 100, 64, 16), (2, 5000, 128, 32), (1, 4096, 512, 32), (4, 333, 256, 32)])
def test_groupnorm_silu(T, HW, C, G):
    from easyanimate_amd import ops
    g = torch.Generator().manual_seed(5)
    x = _bf(torch.randn(T, HW, C, generator=g) * 2 + 0.5).to(DEV)
    gamma = (1 + 0.1 * torch.randn(C, generator=g)).to(DEV)
    beta = (0.1 * torch.randn(C, generator=g)).to(DEV)
    for act in (True, False):
        y = ops.groupnorm_silu(x, gamma, beta, G, 1e-6, act=act)
        ref = F.group_norm(x.double().transpose(1, 2), G, gamma.double(), beta.double(), 1e-6).transpose(1, 2)
        if act:
            ref = F.silu(ref)
        err, rel = _rep(f"groupnorm T{T} HW{HW} C{C} act{int(act)}", y, ref)
        assert rel < 4e-3
    y2 = ops.groupnorm_silu(x, gamma, beta, G, 1e-6, act=True)
    assert torch.equal(y2, ops.groupnorm_silu(x, gamma, beta, G, 1e-6, act=True))  # deterministic (no atomics)


def test_softmax_rows_and_f32_gemm():
    from easyanimate_amd import ops
    g = torch.Generator().manual_seed(6)
    for cols in (64, 1000 - 1000 % 8, 4096, 16384):
        x = (torch.randn(37, cols, generator=g) * 4).to(DEV)
        for xin in (x, x.to(torch.bfloat16)):
            y = ops.softmax_rows(xin.contiguous(), 0.37)
            ref = torch.softmax(xin.double() * 0.37, -1)
            err, rel = _rep(f"softmax {cols} {xin.dtype}", y, ref)
            assert rel < 5e-3
    A = _bf(torch.randn(200, 128, generator=g)).to(DEV)
    W = _bf(torch.randn(72, 128, generator=g)).to(DEV)
    out = ops.gemm(A, W, None, ops.EPI_F32_OUT)
    assert out.dtype == torch.float32
    err, rel = _rep("gemm f32 out", out, A.double() @ W.double().t())
    assert rel < 1e-5


def test_layout_kernels():
    from easyanimate_amd import ops
    g = torch.Generator().manual_seed(7)
    x = torch.randn(3, 4, 5, 6, generator=g).to(DEV)
    for dt in (torch.float32, torch.bfloat16):
        cl = ops.ncdhw_to_ndhwc(x.to(dt), 8)
        assert torch.equal(cl[..., :3], x.to(dt).to(torch.bfloat16).permute(1, 2, 3, 0)) and cl[..., 3:].abs().max() == 0
        back = ops.ndhwc_to_ncdhw(cl, 3, dt)
        assert torch.equal(back, x.to(dt).to(torch.bfloat16).to(dt))
    cl = ops.ncdhw_to_ndhwc(x * 2, 8)
    post = ops.ndhwc_to_ncdhw(cl, 3, torch.float32, post=1)
    ref = ((x * 2).to(torch.bfloat16).float().clamp(-1, 1) / 2 + 0.5).clamp(0, 1)
    assert torch.allclose(post, ref, atol=1e-6)


def _metrics(name, got, ref32, refb):
    got, r, b = got.double().cpu(), ref32.double(), refb.double()
    mse = ((got - r) ** 2).mean().item()
    floor = ((b - r) ** 2).mean().item()
    rel = ((got - r).norm() / r.norm()).item()
    print(f"[parity] {name}: new-bf16 vs ref-fp32 MSE={mse:.3e} rel_l2={rel:.3e} | ref-bf16 floor MSE={floor:.3e} | new vs ref-bf16 MSE={((got - b) ** 2).mean().item():.3e}")
    return mse, floor, rel


def test_vae_vs_golden_chunked_reference():
    from easyanimate_amd import AutoencoderKLMagvit
    from easyanimate_amd.synthetic import synth_state_dict
    g = torch.load(os.path.join(GOLD, "vae_tiny.pt"), weights_only=False)
    vae = AutoencoderKLMagvit.from_config(g["cfg"])
    vae.load_state_dict(synth_state_dict(g["shapes"], g["seed"], g["style"]), strict=True)
    vae = vae.to(torch.bfloat16).to(DEV).eval()
    with torch.no_grad():
        post = vae.encode(g["video"].to(DEV).bfloat16())[0]
        dec = vae.decode(g["z"].to(DEV).bfloat16())[0]
    assert post.mode().shape == (1, 16, 3, 8, 8) and dec.shape == (1, 3, 9, 64, 64)
    mse_e, floor_e, rel_e = _metrics("vae encode moments", post.parameters.float(), g["moments"], g["moments_bf16"])
    mse_d, floor_d, rel_d = _metrics("vae decode", dec.float(), g["dec"], g["dec_bf16"])
    assert mse_e <= max(1e-4, 2 * floor_e) and mse_d <= max(1e-4, 2 * floor_d)
    # fp32 in -> fp32 out
    with torch.no_grad():
        dec32 = vae.decode(g["z"].to(DEV), return_dict=True).sample
    assert dec32.dtype == torch.float32
    # post-processing fused in the layout kernel == decode_latents arithmetic
    with torch.no_grad():
        pp = vae.decode(g["z"].to(DEV), postprocess=True)[0]
    assert torch.allclose(pp, (dec32.clamp(-1, 1) / 2 + 0.5).clamp(0, 1), atol=1e-6)


def test_vae_frame_bookkeeping_gpu():
    from easyanimate_amd import AutoencoderKLMagvit
    from easyanimate_amd.synthetic import synth_state_dict
    g = torch.load(os.path.join(GOLD, "vae_tiny.pt"), weights_only=False)
    vae = AutoencoderKLMagvit.from_config(g["cfg"])
    vae.load_state_dict(synth_state_dict(g["shapes"], g["seed"], g["style"]), strict=True)
    vae = vae.to(torch.bfloat16).to(DEV).eval()
    from oracle import restatement_vae as RV
    sd = synth_state_dict(g["shapes"], g["seed"], g["style"])
    for f, fl in ((1, 1), (5, 2), (13, 4)):
        gen = torch.Generator().manual_seed(f)
        vid = torch.rand(1, 3, f, 64, 64, generator=gen) * 2 - 1
        z = torch.randn(1, 16, fl, 8, 8, generator=gen)
        with torch.no_grad():
            m = vae.encode(vid.to(DEV).bfloat16())[0].parameters
            d = vae.decode(z.to(DEV).bfloat16())[0]
            mr = RV.vae_encode_moments(sd, vid, 16)
            dr = RV.vae_decode(sd, z, 16)
        assert m.shape == mr.shape and d.shape == dr.shape
        _, rel_m = _rep(f"vae enc {f}f", m.float(), mr)
        _, rel_d = _rep(f"vae dec {fl}->{f}f", d.float(), dr)
        assert rel_m < 4e-2 and rel_d < 4e-2


# ---- round 3: temporal nearest x2 kept virtual (upsamplers.py:146-152 never materialised) -----------------------------------
VIRT_CASES = [
    # T_phys, H, W, Ci, Co, forced conv_tile, expected kernel family
    (3, 6, 10, 64, 64, 0, None),                 # 128^2 generic kernel
    (4, 4, 256, 128, 256, 1024, "conv_row16_256"),   # 256-voxel row slab
    (3, 3, 512, 256, 128, 1024, "conv_row16_m512"),  # 512-voxel row slab (decoder block 3, conv1 of the first ResidualBlock3D)
    (3, 20, 24, 128, 128, 256, "conv_pp"),        # ping-pong kernel
    (1, 5, 6, 64, 64, 0, None),                   # one physical frame: nothing to duplicate
]


@pytest.mark.parametrize("T,H,W,Ci,Co,tile,kern", VIRT_CASES)
def test_conv3d_virtual_temporal_duplication(T, H, W, Ci, Co, tile, kern):
    """tdup bits 2 / 4 (virtual input frames / virtual residual frames) against the same convolution on the materialised
    clip: identical arithmetic in the identical order -> bit-identical, in every kernel family the up blocks use."""
    from easyanimate_amd import _lib, ops
    from easyanimate_amd.vae_modules import _pack_conv_weight
    g = torch.Generator().manual_seed(13)
    x = _bf(torch.randn(T, H, W, Ci, generator=g)).to(DEV)
    w = _pack_conv_weight(_bf(torch.randn(Co, Ci, 3, 3, 3, generator=g) / (Ci * 27) ** 0.5)).to(DEV)
    b = torch.randn(Co, generator=g).to(DEV)
    r = _bf(torch.randn(T, H, W, Co, generator=g)).to(DEV)
    idx = ((torch.arange(2 * T - 1, device=DEV) + 1) >> 1) if T > 1 else torch.zeros(1, dtype=torch.long, device=DEV)
    xm, rm = x[idx].contiguous(), r[idx].contiguous()
    _lib.set_option("conv_tile", tile)
    try:
        _lib.reset_counters()
        y_ref = ops.conv3d_cl(xm, w, b, 3, 1, 1, 1, res=rm)
        c_ref = _lib.counters()
        _lib.reset_counters()
        y_v = ops.conv3d_cl(x, w, b, 3, 1, 1, 1, res=r, vin=True, vres=True)
        c_v = _lib.counters()
        y_v2 = ops.conv3d_cl(x, w, b, 3, 1, 1, 1, res=rm, vin=True)          # virtual input, real residual
        y_v3 = ops.conv3d_cl(xm, w, b, 3, 1, 1, 1, res=r, vres=True)          # real input, virtual residual
    finally:
        _lib.set_option("conv_tile", 0)
    torch.cuda.synchronize()
    assert c_ref == c_v and (kern is None or any(k.startswith(kern) for k in c_v)), (c_ref, c_v)
    assert y_v.shape == y_ref.shape == (2 * T - 1 if T > 1 else 1, H, W, Co)
    assert torch.equal(y_v, y_ref) and torch.equal(y_v2, y_ref) and torch.equal(y_v3, y_ref)
    # the GroupNorm partial sums of the epilogue ride along unchanged
    pv, pr = getattr(y_v, "gn_partial", None), getattr(y_ref, "gn_partial", None)
    assert (pv is None) == (pr is None)
    if pv is not None:
        n = y_ref.shape[0] * pr[1] * (Co // 4) * 2
        assert pv[1] == pr[1] and torch.equal(pv[0][:n], pr[0][:n])


def test_vae_decode_virtual_equals_materialised_temporal_duplication():
    """The decoder with the up-samplers' duplicated frames kept virtual (GroupNorm, 1x1x1 shortcut, first 3x3x3 convolution
    and the residual add of the next block address frame (t+1)>>1) against the decoder that writes them: every kernel does
    the same arithmetic on the same values -> the decoded clips are bit-identical."""
    from easyanimate_amd import AutoencoderKLMagvit, vae_modules
    from easyanimate_amd.synthetic import synth_state_dict
    g = torch.load(os.path.join(GOLD, "vae_tiny.pt"), weights_only=False)
    vae = AutoencoderKLMagvit.from_config(g["cfg"])
    vae.load_state_dict(synth_state_dict(g["shapes"], g["seed"], g["style"]), strict=True)
    vae = vae.to(torch.bfloat16).to(DEV).eval()
    gen = torch.Generator().manual_seed(4)
    for fl, hw in ((3, 8), (4, 16), (1, 8)):
        z = torch.randn(1, 16, fl, hw, hw, generator=gen).to(DEV).bfloat16()
        with torch.no_grad():
            assert vae_modules.VIRTUAL_TDUP
            a = vae.decode(z)[0]
            vae_modules.VIRTUAL_TDUP = False
            try:
                b = vae.decode(z)[0]
            finally:
                vae_modules.VIRTUAL_TDUP = True
        assert a.shape == b.shape == (1, 3, 4 * (fl - 1) + 1, 8 * hw, 8 * hw) and torch.equal(a, b)


def test_conv3d_narrow_n_chunked_equals_whole_clip():
    """conv3d_narrow bounds its fp32 tap-plane scratch by processing frame chunks (each with the two frames in front of it):
    bit-identical to the whole-clip evaluation, first chunk (causal replicate padding) included."""
    from easyanimate_amd import ops
    g = torch.Generator().manual_seed(21)
    T, H, W, Ci, Co = 11, 16, 24, 128, 3
    x = _bf(torch.randn(T, H, W, Ci, generator=g)).to(DEV)
    wz = _bf(torch.randn(27 * Co, Ci, generator=g) / (27 * Ci) ** 0.5).to(DEV)
    b = torch.randn(Co, generator=g).to(DEV)
    whole = ops.conv3d_narrow(x, wz, b, Co, 8)
    lim = ops.NARROW_SCRATCH_BYTES
    ops.NARROW_SCRATCH_BYTES = 6 * H * W * 27 * Co * 4        # chunks of four frames (+ two in front)
    try:
        chunked = ops.conv3d_narrow(x, wz, b, Co, 8)
    finally:
        ops.NARROW_SCRATCH_BYTES = lim
    assert torch.equal(chunked, whole)


@pytest.mark.parametrize("T,n_q,n_keys,amp", [(2, 300, 300, 1.0), (1, 1024, 1024, 1.0), (3, 512, 1024, 1.0), (1, 200, 5400, 1.0),
                                                (2, 640, 640, 6.0)])
def test_attention_head_dim_512_flash(T, n_q, n_keys, amp):
    """ea_attention_d512_fwd_bf16 (the VAE mid block's single 512-channel head, vaemodules/attention.py:391-423) against an
    fp64 softmax(Q K^T / sqrt(512)) V: query / key tails, queries != keys (the spatially split VAE), and -- amp = 6 -- score
    ranges that move the lazy softmax shift many times (scores spread over +-100 in log2 units)."""
    from easyanimate_amd import _lib, ops
    g = torch.Generator().manual_seed(17 + n_q)
    n_pad = ops.round_up(n_keys, 64)
    q = _bf(torch.randn(T, n_q, 512, generator=g) * amp)
    k = torch.zeros(T, n_pad, 512, dtype=torch.bfloat16)
    k[:, :n_keys] = _bf(torch.randn(T, n_keys, 512, generator=g) * amp)
    k[:, n_keys:] = 50.0                                   # padded key rows hold garbage: they must be masked, not attended
    v = _bf(torch.randn(T, n_keys, 512, generator=g))
    vt = torch.zeros(T, 512, n_pad, dtype=torch.bfloat16)
    vt[:, :, :n_keys] = v.transpose(1, 2)
    ref = torch.softmax(q.double() @ k[:, :n_keys].double().transpose(1, 2) * 512 ** -0.5, -1) @ v.double()
    _lib.reset_counters()
    out = ops.attention_d512(q.to(DEV), k.to(DEV), vt.to(DEV), n_keys, 512 ** -0.5)
    out2 = ops.attention_d512(q.to(DEV), k.to(DEV), vt.to(DEV), n_keys, 512 ** -0.5)
    torch.cuda.synchronize()
    assert _lib.counters() == {"attention_d512": 2} and torch.equal(out, out2)
    err, rel = _rep(f"attention head_dim 512 T{T} q{n_q} k{n_keys} amp{amp:g}", out, ref)
    assert rel < 8e-3


def test_mid_block_attention_flash_equals_three_gemm_route():
    """SpatialAttention at full width (512 channels) through the flash kernel against the Q K^T GEMM -> fp32 logits -> row
    softmax -> P V GEMM route it replaces (both against fp64 elsewhere): same projections, same residual; the two differ
    only in where the probabilities are rounded (normalised bf16 P vs bf16 exp with one division at the end)."""
    from easyanimate_amd import _lib, vae_modules
    from easyanimate_amd.vae_modules import SpatialAttention
    torch.manual_seed(3)
    att = SpatialAttention(512, nheads=1, head_dim=512).to(torch.bfloat16).to(DEV).eval()
    with torch.no_grad():
        for p_ in att.parameters():
            if p_.dim() == 2:
                p_.copy_(torch.randn_like(p_.float()) * (1.5 / p_.shape[1] ** 0.5))
    x = torch.randn(2, 20, 24, 512, device=DEV).to(torch.bfloat16)     # n = 480 tokens: padded keys
    with torch.no_grad():
        _lib.reset_counters()
        a = att(x)
        ca = _lib.counters()
        vae_modules.FLASH_MID_BLOCK = False
        try:
            _lib.reset_counters()
            b = att(x)
            cb = _lib.counters()
        finally:
            vae_modules.FLASH_MID_BLOCK = True
    assert ca.get("attention_d512", 0) == 1 and "attention_d512" not in cb
    d = (a.float() - b.float()).abs()
    rel = (d.norm() / b.float().norm()).item()
    print(f"[parity] mid-block attention flash vs three-GEMM route: rel_l2 {rel:.3e}, max {d.max().item():.3e}")
    assert rel < 4e-3


@pytest.mark.parametrize("T,H,W,Ci,Co,tdup", [(3, 4, 256, 64, 256, False), (2, 3, 512, 128, 256, True), (1, 2, 256, 256, 512, True),
                                               (2, 5, 256, 512, 512, False)])
def test_conv3d_subpixel_upsample(T, H, W, Ci, Co, tdup):
    """ea_conv3d_cl_subpixel_bf16: "nearest x2, then 3x3x3 causal convolution" as four 12-tap parity classes on the source grid,
    against the fp64 convolution of the up-sampled clip (borders: the zero padding of the UP-SAMPLED image; frame 0: causal
    replicate; tdup: duplicate store) and against the 27-tap kernel with the x2 folded into its addressing (they differ by
    the bf16 rounding of the summed weights only).  The GroupNorm partial sums of the epilogue must give the statistics of
    the output."""
    from easyanimate_amd import _lib, ops
    from easyanimate_amd.vae_modules import _pack_conv_weight, _pack_subpixel_weight
    g = torch.Generator().manual_seed(5 + W + Ci)
    x = _bf(torch.randn(1, Ci, T, H, W, generator=g))
    w = _bf(torch.randn(Co, Ci, 3, 3, 3, generator=g) / (Ci * 27) ** 0.5)
    b = torch.randn(Co, generator=g)
    xr = F.interpolate(x.double(), scale_factor=(1, 2, 2), mode="nearest")
    xr = F.pad(xr, (0, 0, 0, 0, 2, 0), mode="replicate")
    ref = F.conv3d(xr, w.double(), b.double(), padding=(0, 1, 1))
    if tdup and T > 1:
        ref = torch.cat([ref[:, :, :1], F.interpolate(ref[:, :, 1:], scale_factor=(2, 1, 1), mode="nearest")], 2)
    xcl = x[0].permute(1, 2, 3, 0).contiguous().to(DEV)
    _lib.reset_counters()
    y = ops.conv3d_subpixel(xcl, _pack_subpixel_weight(w).to(DEV), b.to(DEV), tdup=tdup)
    assert _lib.counters() == {"conv_row16_256_subpixel": 1}
    y2 = ops.conv3d_subpixel(xcl, _pack_subpixel_weight(w).to(DEV), b.to(DEV), tdup=tdup)
    assert torch.equal(y, y2)
    got = y.permute(3, 0, 1, 2)[None]
    assert got.shape == ref.shape
    err, rel = _rep(f"sub-pixel up-sampling conv T{T} {H}x{W} {Ci}->{Co} tdup{int(tdup)}", got, ref)
    assert rel < 5e-3
    y27 = ops.conv3d_cl(xcl, _pack_conv_weight(w).to(DEV), b.to(DEV), 3, 1, 1, 1, ups=True, tdup=tdup)
    d = (y.float() - y27.float())
    rel27 = (d.norm() / y27.float().norm()).item()
    print(f"[parity] sub-pixel vs 27-tap folded kernel: rel_l2 {rel27:.3e}")
    assert y27.shape == y.shape and rel27 < 5e-3
    # fused GroupNorm statistics == statistics of the tensor
    gam, bet = torch.ones(Co, device=DEV), torch.zeros(Co, device=DEV)
    assert getattr(y, "gn_partial", None) is not None
    n1 = ops.groupnorm_silu(y, gam, bet, 32, 1e-6, act=False)
    ops.FUSED_GN_STATS = False
    try:
        n2 = ops.groupnorm_silu(y, gam, bet, 32, 1e-6, act=False)
    finally:
        ops.FUSED_GN_STATS = True
    assert (n1.float() - n2.float()).abs().max().item() <= 2 ** -6


@pytest.mark.parametrize("T,H,W,Ci,Co,kern", [(4, 4, 256, 128, 256, "conv_row16_256"), (3, 3, 512, 256, 128, "conv_row16_m512"),
                                               (2, 2, 512, 64, 128, "conv_row16_m512"), (3, 2, 256, 64, 128, "conv_row16_128")])
def test_conv3d_merged_temporal_taps(T, H, W, Ci, Co, kern):
    """tdup bit 8: a 3x3x3 layer behind a virtual temporal x2 with its temporal taps merged onto the two physical frames it
    touches (18 taps), against the fp64 convolution of the materialised clip and against the 27-tap kernel with the frame map
    in its addressing (they differ by the bf16 rounding of the summed weights only); with a virtual residual; frame 0 (all
    three taps on one frame) included."""
    from easyanimate_amd import _lib, ops
    from easyanimate_amd.vae_modules import _pack_conv_weight, _pack_tmerge_weight
    g = torch.Generator().manual_seed(41 + W + Ci)
    u = _bf(torch.randn(T, H, W, Ci, generator=g))
    w = _bf(torch.randn(Co, Ci, 3, 3, 3, generator=g) / (Ci * 27) ** 0.5)
    b = torch.randn(Co, generator=g)
    r = _bf(torch.randn(T, H, W, Co, generator=g))
    idx = (torch.arange(2 * T - 1) + 1) >> 1
    xm = u[idx].permute(3, 0, 1, 2)[None].double()
    ref = F.conv3d(F.pad(xm, (0, 0, 0, 0, 2, 0), mode="replicate"), w.double(), b.double(), padding=(0, 1, 1))
    ref = ref + r[idx].permute(3, 0, 1, 2)[None].double()
    _lib.set_option("conv_tile", 1024)
    try:
        assert ops.conv3d_tmerge_ok(2 * T - 1, H, W, Ci, Co)
        _lib.reset_counters()
        y = ops.conv3d_cl(u.to(DEV), _pack_tmerge_weight(w, Co).to(DEV), b.to(DEV), 3, 1, 1, 1, res=r.to(DEV), vin=True, vres=True, tmerge=True)
        c = _lib.counters()
        y2 = ops.conv3d_cl(u.to(DEV), _pack_tmerge_weight(w, Co).to(DEV), b.to(DEV), 3, 1, 1, 1, res=r.to(DEV), vin=True, vres=True, tmerge=True)
        y27 = ops.conv3d_cl(u.to(DEV), _pack_conv_weight(w).to(DEV), b.to(DEV), 3, 1, 1, 1, res=r.to(DEV), vin=True, vres=True)
    finally:
        _lib.set_option("conv_tile", 0)
    torch.cuda.synchronize()
    assert sorted(c) == sorted([kern, "conv_tmerge_18_taps"]) and c["conv_tmerge_18_taps"] == 1 and torch.equal(y, y2), c
    got = y.permute(3, 0, 1, 2)[None]
    assert got.shape == ref.shape
    err, rel = _rep(f"merged temporal taps T{T} {H}x{W} {Ci}->{Co}", got, ref)
    rel27 = ((y.float() - y27.float()).norm() / y27.float().norm()).item()
    print(f"[parity] merged-tap vs 27-tap kernel: rel_l2 {rel27:.3e}")
    assert rel < 5e-3 and rel27 < 5e-3


# ---- round 5: the row-slab kernels on four waves with the hand-placed main loop (conv3d_cl_row16_w4a_kernel) -------------------------
@pytest.mark.parametrize("T,H,W,Ci,Co,res,tmerge", [
    (3, 3, 512, 128, 128, True, False),       # 512 x 128 tiles: the kernel that carries 42 % of a 49 x 1024^2 decode
    (2, 2, 1024, 256, 128, True, False),      # two tiles per row, 8 channel blocks
    (2, 5, 512, 64, 128, False, False),       # 2 channel blocks: the (dt, dh) cursor moves every second slab; rows of padding top and bottom
    (3, 3, 512, 256, 128, True, True),        # merged temporal taps (6 table rows), virtual input and residual
    (2, 2, 512, 64, 128, False, True),
    (4, 4, 256, 128, 256, True, False),       # 256 x 256 tiles
    (2, 3, 512, 256, 256, False, False),
    (3, 2, 256, 512, 512, True, False),       # two N tiles
])
def test_conv_w4a_bit_identical_to_the_eight_wave_kernels(T, H, W, Ci, Co, res, tmerge):
    """ea_set_option("conv_w4a", 3): same slabs, same fragments, the same MFMAs accumulating the same taps in the same order per
    accumulator -> the output and the GroupNorm partial sums are bit-identical to conv3d_cl_row16_k32_kernel (which is pinned to the
    fp64 convolution by test_conv3d_cl_row_slab / test_conv3d_merged_temporal_taps)."""
    from easyanimate_amd import _lib, ops
    from easyanimate_amd.vae_modules import _pack_conv_weight, _pack_tmerge_weight
    g = torch.Generator().manual_seed(77 + W + Ci + Co)
    x = _bf(torch.randn(T, H, W, Ci, generator=g)).to(DEV)
    w = _bf(torch.randn(Co, Ci, 3, 3, 3, generator=g) / (Ci * 27) ** 0.5)
    b = torch.randn(Co, generator=g).to(DEV)
    wp = (_pack_tmerge_weight(w, Co) if tmerge else _pack_conv_weight(w)).to(DEV)
    Tl = 2 * T - 1 if tmerge else T
    r = _bf(torch.randn(T if tmerge else Tl, H, W, Co, generator=g)).to(DEV) if res else None
    kw = dict(vin=True, vres=res, tmerge=True) if tmerge else {}
    outs = {}
    _lib.set_option("conv_tile", 1024)
    _lib.set_option("conv_m512", 3)            # the eight-wave reference for the 256 x 256 tiles is the k32 kernel too
    try:
        for v in (0, 3, 3):
            _lib.set_option("conv_w4a", v)
            _lib.reset_counters()
            y = ops.conv3d_cl(x, wp, b, 3, 1, 1, 1, res=r, **kw)
            torch.cuda.synchronize()
            c = _lib.counters()
            assert c.get("conv_w4a", 0) == (1 if v else 0) and (c.get("conv_row16_m512", 0) + c.get("conv_row16_256_k32", 0)) == 1, c
            part, nblk = y.gn_partial
            outs.setdefault(v, []).append((y.clone(), part[:y.shape[0] * nblk * (Co // 4) * 2].clone(), nblk))
    finally:
        _lib.set_option("conv_tile", 0)
        _lib.set_option("conv_m512", 1)
        _lib.set_option("conv_w4a", 0)         # (the module fixture restores the default)
    (y0, p0, n0), (y1, p1, n1), (y2, p2, n2) = outs[0][0], outs[3][0], outs[3][1]
    assert torch.isfinite(y1.float()).all() and n0 == n1
    if not torch.equal(y1, y0):
        d = (y1.float() - y0.float()).abs()
        bad = (d > 0).nonzero()
        raise AssertionError(f"w4a conv differs: {bad.shape[0]} of {d.numel()} elements, max {d.max().item():.3e}; first {bad[:4].tolist()} last {bad[-2:].tolist()}; "
                             f"frames {bad[:, 0].unique().tolist()} rows {bad[:, 1].unique().tolist()[:8]} voxels {bad[:, 2].unique().tolist()[:16]} channels {bad[:, 3].unique().tolist()[:16]}")
    assert torch.equal(p1, p0)
    assert torch.equal(y2, y1) and torch.equal(p2, p1)       # repeated launches agree (race screen)


# ---- round 6: the wrapper's spatial tiling (autoencoder_magvit.py:249-254,276-279,319-448) --------------------------------------------
def _ref_blend_v(a, b, e):     # autoencoder_magvit.py:319-327, verbatim arithmetic on fp64 copies
    e = min(a.shape[3], b.shape[3], e)
    for y in range(e):
        b[:, :, :, y, :] = a[:, :, :, -e + y, :] * (1 - y / e) + b[:, :, :, y, :] * (y / e)
    return b


def _ref_blend_h(a, b, e):     # :329-337
    e = min(a.shape[4], b.shape[4], e)
    for x in range(e):
        b[:, :, :, :, x] = a[:, :, :, :, -e + x] * (1 - x / e) + b[:, :, :, :, x] * (x / e)
    return b


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("ha,hb,wa,wb,extent", [(48, 48, 48, 48, 12), (48, 28, 48, 20, 12), (7, 5, 9, 3, 12), (384, 224, 384, 160, 96)])
def test_tile_blend_kernels(dtype, ha, hb, wa, wb, extent):
    """ea_tile_blend / ea_tile_corner_blend against the reference's blend_v / blend_h / corner arithmetic in fp64 (tiles of different
    sizes, an extent larger than a tile)."""
    from easyanimate_amd import ops
    g = torch.Generator().manual_seed(ha * 7 + wb)
    B, C, T = 1, 3, 2
    tol = 1e-6 if dtype == torch.float32 else 3e-2      # one bf16 rounding of values up to ~5: half an ulp = 2^-7
    a, b = torch.randn(B, C, T, ha, wa, generator=g).to(dtype), torch.randn(B, C, T, hb, wa, generator=g).to(dtype)
    got = ops.tile_blend_(a.to(DEV), b.clone().to(DEV), extent, 3)
    ref = _ref_blend_v(a.double(), b.double().clone(), extent)
    assert (got.double().cpu() - ref).abs().max().item() < tol
    a, b = torch.randn(B, C, T, ha, wa, generator=g).to(dtype), torch.randn(B, C, T, ha, wb, generator=g).to(dtype)
    got = ops.tile_blend_(a.to(DEV), b.clone().to(DEV), extent, 4)
    ref = _ref_blend_h(a.double(), b.double().clone(), extent)
    assert (got.double().cpu() - ref).abs().max().item() < tol
    # the corner: weights = min(linspace_x, linspace_y) (:432-445)
    dec, q = torch.randn(B, C, T, ha + 5, wa + 3, generator=g).to(dtype), torch.randn(B, C, T, hb, wb, generator=g).to(dtype)
    got = ops.tile_corner_blend_(q.to(DEV), dec.clone().to(DEV))
    wts = torch.min(torch.linspace(0, 1, wb, dtype=torch.float64).unsqueeze(0).repeat(hb, 1),
                    torch.linspace(0, 1, hb, dtype=torch.float64).unsqueeze(1).repeat(1, wb))
    ref = dec.double().clone()
    ref[:, :, :, -hb:, -wb:] = wts * q.double() + (1 - wts) * ref[:, :, :, -hb:, -wb:]
    assert (got.double().cpu() - ref).abs().max().item() < tol


def test_vae_tiling_vs_reference_golden():
    """use_tiling=True through the product wrapper -- tiled_encode / tiled_decode as host loops over the whole-tile kernels, seams by
    ea_tile_blend, the corner by ea_tile_corner_blend -- against the UNCHANGED reference run with use_tiling=True (full width,
    5 x 512 x 448, tile 384 / overlap 0.25: 2 x 2 tiles of unequal sizes; oracle/gen_golden.py section vae_tiled).  Also: the tiled
    result really is the tiled one (it differs from the untiled result about as much as the reference's does), the per-component
    switches, and upcast_vae's refusal."""
    from easyanimate_amd import AutoencoderKLMagvit
    from easyanimate_amd.synthetic import synth_state_dict
    from oracle.gen_golden import vae_tiled_inputs
    g = torch.load(os.path.join(GOLD, "vae_tiled_5x512x448.pt"), weights_only=False)
    video, z = vae_tiled_inputs(g["input_seed"], g["frames"], g["height"], g["width"])
    assert abs(video.double().sum().item() - g["video_sum"]) < 1e-6 and abs(z.double().sum().item() - g["z_sum"]) < 1e-6
    with torch.device("meta"):
        vae = AutoencoderKLMagvit.from_config(g["cfg"])
    assert vae.use_tiling and vae.tile_latent_min_size == 48
    shapes = {k: tuple(v.shape) for k, v in vae.state_dict().items()}
    vae = vae.to_empty(device="cpu")
    vae.load_state_dict(synth_state_dict(shapes, g["seed"], g["style"]), strict=True)
    vae = vae.to(torch.bfloat16).to(DEV).eval()
    with torch.no_grad():
        mom = vae.encode(video.to(DEV).bfloat16())[0].parameters.float().cpu()
        dec = vae.decode(z.to(DEV).bfloat16())[0].float().cpu()
        vae.use_tiling = False
        mom_u = vae.encode(video.to(DEV).bfloat16())[0].parameters.float().cpu()
        dec_u = vae.decode(z.to(DEV).bfloat16())[0].float().cpu()
        vae.use_tiling_decoder = True                      # the per-component switches (:252-254, 278-279)
        dec_d = vae.decode(z.to(DEV).bfloat16())[0].float().cpu()
        mom_d = vae.encode(video.to(DEV).bfloat16())[0].parameters.float().cpu()
    assert mom.shape == g["moments"].shape == (1, 32, 2, 64, 56) and dec.shape == g["dec"].shape == (1, 3, 5, 512, 448)
    mse = lambda a_, b_: ((a_.double() - b_.double()) ** 2).mean().item()
    m_e, m_d = mse(mom, g["moments"].float()), mse(dec, g["dec"].float())
    u_e, u_d = mse(mom, mom_u), mse(dec, dec_u)
    print(f"[parity] tiled VAE (5 x 512 x 448, tile 384 / 0.25) vs the reference's tiled fp32 run: moments MSE {m_e:.3e}, decode MSE {m_d:.3e} "
          f"(bar 1e-4); tiled-vs-untiled here {u_e:.3e} / {u_d:.3e}, in the reference {g['untiled_mse'][0]:.3e} / {g['untiled_mse'][1]:.3e}")
    assert m_e < 1e-4 and m_d < 1e-4
    assert 0.3 * g["untiled_mse"][0] < u_e < 3 * g["untiled_mse"][0] + 1e-4 and 0.3 * g["untiled_mse"][1] < u_d < 3 * g["untiled_mse"][1] + 1e-4
    assert torch.equal(dec_d, dec) and torch.equal(mom_d, mom_u)
    with pytest.raises(NotImplementedError, match="upcast_vae"):
        AutoencoderKLMagvit.from_config(dict(g["cfg"], upcast_vae=True))


# ---- round 6: channel-blocked GroupNorm output read by the four-wave row-slab convolutions --------------------------------------------
@pytest.mark.parametrize("T,H,W,C,act", [(3, 24, 256, 128, True), (2, 16, 256, 256, True), (2, 9, 256, 512, False), (1, 5, 72, 128, True)])
def test_groupnorm_blocked_output_is_the_permuted_output(T, H, W, C, act):
    """ea_groupnorm_apply_bf16 with act bit 1: the same values, written [C / 32][T][H][W][32]."""
    from easyanimate_amd import ops
    g = torch.Generator().manual_seed(C + T)
    x = _bf(torch.randn(T, H, W, C, generator=g) * 2 + 0.3).to(DEV)
    gamma, beta = (1 + 0.1 * torch.randn(C, generator=g)).to(DEV), (0.1 * torch.randn(C, generator=g)).to(DEV)
    y = ops.groupnorm_silu(x, gamma, beta, 32, 1e-6, act=act)
    yb = ops.groupnorm_silu(x, gamma, beta, 32, 1e-6, act=act, blocked=True)
    assert yb.shape == (C // 32, T, H, W, 32)
    assert torch.equal(yb, y.view(T, H, W, C // 32, 32).permute(3, 0, 1, 2, 4).contiguous())


@pytest.mark.parametrize("T,H,W,Ci,Co,res,virt", [(3, 3, 512, 128, 128, True, False), (4, 4, 256, 128, 256, True, False), (3, 2, 256, 512, 512, False, False),
                                                  (3, 3, 512, 256, 128, True, True), (2, 4, 256, 256, 256, False, True)])
def test_conv_on_blocked_input_bit_identical(T, H, W, Ci, Co, res, virt):
    """ops.conv3d_cl(..., blocked=True) (tdup bit 4; conv3d_cl_row16_w4a_kernel<.., .., true>): the slabs come in as 1 KiB pieces of
    the channel-blocked tensor -- same values into the same LDS rows: output and GroupNorm partial sums bit-identical to the
    voxel-major input, with a residual, with a virtually duplicated input and merged temporal taps."""
    from easyanimate_amd import _lib, ops
    from easyanimate_amd.vae_modules import _pack_conv_weight, _pack_tmerge_weight
    g = torch.Generator().manual_seed(91 + W + Ci + Co)
    x = _bf(torch.randn(T, H, W, Ci, generator=g)).to(DEV)
    xb = x.view(T, H, W, Ci // 32, 32).permute(3, 0, 1, 2, 4).contiguous()
    w = _bf(torch.randn(Co, Ci, 3, 3, 3, generator=g) / (Ci * 27) ** 0.5)
    b = torch.randn(Co, generator=g).to(DEV)
    wp = (_pack_tmerge_weight(w, Co) if virt else _pack_conv_weight(w)).to(DEV)
    Tl = 2 * T - 1 if virt else T
    r = _bf(torch.randn(Tl, H, W, Co, generator=g)).to(DEV) if res else None
    prev = _lib.get_option("conv_tile"), _lib.get_option("conv_w4a")
    try:
        _lib.set_option("conv_tile", 1024)       # (the shapes are too small for the automatic choice of the row-slab kernels)
        _lib.set_option("conv_w4a", 3)
        assert ops.conv3d_blocked_ok(Tl, H, W, Ci, Co)
        kw = dict(res=r, vin=virt, tmerge=virt)
        _lib.reset_counters()
        y0 = ops.conv3d_cl(x, wp, b, 3, **kw)
        y1 = ops.conv3d_cl(xb, wp, b, 3, blocked=True, **kw)
        torch.cuda.synchronize()
        c = _lib.counters()
        assert c.get("conv_blocked_input", 0) == 1 and c.get("conv_w4a", 0) == 2, c
        assert torch.equal(y0, y1) and y0.shape == (Tl, H, W, Co)
        assert y0.gn_partial[1] == y1.gn_partial[1] and torch.equal(y0.gn_partial[0][:y0.gn_partial[1] * Tl * (Co // 4) * 2], y1.gn_partial[0][:y1.gn_partial[1] * Tl * (Co // 4) * 2])
        _lib.set_option("conv_w4a", 0)
        assert not ops.conv3d_blocked_ok(Tl, H, W, Ci, Co)
        with pytest.raises(RuntimeError, match="channel-blocked"):
            ops.conv3d_cl(xb, wp, b, 3, blocked=True, **kw)
    finally:
        _lib.set_option("conv_tile", prev[0])
        _lib.set_option("conv_w4a", prev[1])


def test_vae_blocked_groupnorm_outputs_change_no_bit():
    """vae_modules.BLOCKED_GN_OUTPUT: the decode and encode of a clip wide enough for the four-wave row-slab kernels (5 x 512 x 512)
    with and without the channel-blocked GroupNorm -> convolution edges: identical bits, and the blocked path really ran."""
    from easyanimate_amd import AutoencoderKLMagvit, _lib, vae_modules
    from easyanimate_amd.synthetic import synth_state_dict
    g = torch.load(os.path.join(GOLD, "vae_dec_5x512.pt"), weights_only=False)
    with torch.device("meta"):
        vae = AutoencoderKLMagvit.from_config(g["cfg"])
    shapes = {k: tuple(v.shape) for k, v in vae.state_dict().items()}
    vae = vae.to_empty(device="cpu")
    vae.load_state_dict(synth_state_dict(shapes, g["seed"], g["style"]), strict=True)
    vae = vae.to(torch.bfloat16).to(DEV).eval()
    gen = torch.Generator().manual_seed(5)
    z = torch.randn(1, 16, 2, 64, 64, generator=gen).to(DEV).bfloat16()
    video = (torch.rand(1, 3, 5, 512, 512, generator=gen) * 2 - 1).to(DEV).bfloat16()
    outs = {}
    prev = vae_modules.BLOCKED_GN_OUTPUT
    prev_w4a = _lib.get_option("conv_w4a")
    try:
        _lib.set_option("conv_w4a", 3)            # (this file's fixture pins the eight-wave kernels; the blocked input is the four-wave kernels')
        for flag in (False, True):
            vae_modules.BLOCKED_GN_OUTPUT = flag
            _lib.reset_counters()
            with torch.no_grad():
                outs[flag] = (vae.decode(z)[0], vae.encode(video)[0].parameters, _lib.counters().get("conv_blocked_input", 0))
    finally:
        vae_modules.BLOCKED_GN_OUTPUT = prev
        _lib.set_option("conv_w4a", prev_w4a)
    print(f"[dispatch] convolutions on a channel-blocked input in one 5 x 512^2 decode + encode: {outs[True][2]} (switch off: {outs[False][2]})")
    assert outs[False][2] == 0 and outs[True][2] >= 8
    assert torch.equal(outs[False][0], outs[True][0]) and torch.equal(outs[False][1], outs[True][1])
