"""Where does a 256^2 GEMM workgroup spend its time?  Needs a diagnostic build:
    EA_HIPCC_EXTRA=-DEA_GEMM_TIMESTAMPS python -m easyanimate_amd.build --force
prints per-workgroup s_memtime deltas: prologue (start -> first tile landed), main loop, epilogue."""
import ctypes
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from easyanimate_amd import _lib, ops

lib = _lib.load()
_lib.set_option("gemm_tile", 256)
print(json.dumps({"gemm_mfma": _lib.get_option("gemm_mfma")}))
for (M, N, K, epi) in [(106496, 3072, 3072, 0), (106496, 12288, 3072, 1), (106496, 3072, 12288, 2), (8192, 8192, 8192, 0)]:
    A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    W = (torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda")
    o = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    gate = torch.randn(1, N, device="cuda")
    run = (lambda: ops.gemm(A, W, bias, 2, out=o, res=o, gate=gate)) if epi == 2 else (lambda: ops.gemm(A, W, bias, epi, out=o))
    run(); run()
    tiles_m, tiles_n = (M + 255) // 256, N // 256
    rpx = (tiles_m + 7) // 8 if tiles_m >= 64 else 0
    nblk = 8 * rpx * tiles_n if rpx else tiles_m * tiles_n
    ts = torch.zeros(nblk * 5, dtype=torch.int64, device="cuda")
    lib.ea_debug_gemm_timestamps.argtypes = [ctypes.c_void_p]
    assert lib.ea_debug_gemm_timestamps(ts.data_ptr()) == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(); e1.record(); torch.cuda.synchronize()
    lib.ea_debug_gemm_timestamps(None)
    t = ts.view(nblk, 5).cpu()
    t = t[t[:, 3] != 0].double()
    pro, loop, epi_t, tot = t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2], t[:, 3] - t[:, 0]
    span = (t[:, 3].max() - t[:, 0].min()).item()
    ms = e0.elapsed_time(e1)
    tick_ns = ms * 1e6 / span
    med = lambda x: x.median().item()
    print(json.dumps({"M": M, "N": N, "K": K, "epi": epi, "kernel_ms": ms, "wgs": int(t.shape[0]), "tick_ns(upper bound)": tick_ns,
                      "ticks_per_wg": {"prologue": med(pro), "mainloop": med(loop), "epilogue": med(epi_t), "total": med(tot)},
                      "mainloop_ticks_per_ktile": med(loop) / (K // 64),
                      "sum_wg_ticks / (256 CUs * span)": tot.sum().item() / (256 * span)}), flush=True)
    del A, W, o
