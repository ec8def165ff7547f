"""Where does a 256^2 GEMM workgroup spend its time?  Needs a diagnostic build beside the product library:
    EA_HIPCC_EXTRA=-DEA_GEMM_TIMESTAMPS EA_LIB_OUT=easyanimate_amd/lib/diag/libea_diag.so python -m easyanimate_amd.build
    EA_LIB_PATH=easyanimate_amd/lib/diag/libea_diag.so python tools/gemm_anatomy.py
Per output tile, s_memtime stamps of wave 0: start -> main asm entered (set-up), the main asm (first requests .. last MFMA), the
epilogue up to its last store ISSUED, and until those stores are acknowledged.  First argument: the gemm_w4a option values to run
(3 = gemm256_w4a_kernel, the product; 0 = the eight-wave kernel, which stamps start / first tile landed / loop end / end).
__builtin_readcyclecounter counts shader-clock cycles (about 1.85 GHz under load), every XCD from its own origin: only differences
inside a workgroup are used."""
import ctypes
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from easyanimate_amd import _lib, ops

lib = _lib.load()
_lib.set_option("gemm_tile", 256)
lib.ea_debug_gemm_timestamps.argtypes = [ctypes.c_void_p]
variants = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [3]
# optional second argument: stagger values to sweep (diagnostic: the first workgroup of a CU starts ((cu >> 3) & 3) * n * 8128 cycles late,
# which puts the CUs of an XCD in four phases -- do the epilogues' stores go faster when they do not all come at once?)
staggers = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0]
lib.ea_debug_gemm_stagger.argtypes = [ctypes.c_int]
for (M, N, K, epi) in [(106496, 12288, 3072, 1), (106496, 9216, 3072, 0), (106496, 3072, 3072, 2), (106496, 3072, 12288, 2), (8192, 8192, 8192, 0)]:
    A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    W = (torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda")
    o = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    gate = torch.randn(1, N, device="cuda")
    run = (lambda: ops.gemm(A, W, bias, 2, out=o, res=o, gate=gate)) if epi == 2 else (lambda: ops.gemm(A, W, bias, epi, out=o))
    tiles_m, tiles_n = (M + 255) // 256, N // 256
    rpx = (tiles_m + 7) // 8 if tiles_m >= 64 else 0
    nblk = 8 * rpx * tiles_n if rpx else tiles_m * tiles_n
    for v, stg in [(v, g) for v in variants for g in staggers]:
        _lib.set_option("gemm_w4a", v)
        assert lib.ea_debug_gemm_stagger(stg) == 0
        run(); run()
        ts = torch.zeros(nblk * 5, dtype=torch.int64, device="cuda")
        assert lib.ea_debug_gemm_timestamps(ts.data_ptr()) == 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(); e1.record(); torch.cuda.synchronize()
        lib.ea_debug_gemm_timestamps(None)
        t = ts.view(nblk, 5).cpu()
        t = t[t[:, 3] != 0].double()
        ms = e0.elapsed_time(e1)
        med = lambda x: int(x.median().item())
        tot = t[:, 4] - t[:, 0]
        rec = {"M": M, "N": N, "K": K, "epi": epi, "gemm_w4a": v, "stagger": stg, "kernel_ms": round(ms, 4), "tiles": int(t.shape[0]),
               "median_cycles_per_tile": {"set-up": med(t[:, 1] - t[:, 0]), "main asm": med(t[:, 2] - t[:, 1]), "epilogue (stores issued)": med(t[:, 3] - t[:, 2]),
                                          "store drain": med(t[:, 4] - t[:, 3]), "total": med(tot)},
               "main_asm_cycles_per_ktile": round((t[:, 2] - t[:, 1]).median().item() / (K // 64), 1),
               # every CU runs its tiles back to back: cycles of all tiles / 256 CUs / kernel time = the clock the stamps ran at (lower bound)
               "implied_GHz": round(tot.sum().item() / 256 / (ms * 1e6), 3)}
        print(json.dumps(rec), flush=True)
    _lib.set_option("gemm_w4a", 3)
    lib.ea_debug_gemm_stagger(0)
    del A, W, o
