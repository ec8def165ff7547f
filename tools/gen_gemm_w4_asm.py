"""Generates easyanimate_amd/csrc/ea_gemm_w4_loop.inc: the hand-placed main loop of gemm256_w4a_kernel (ea_gemm.hip) as ONE
inline-asm block, plus the accumulator read-out statements.

    python tools/gen_gemm_w4_asm.py            (rewrites the .inc; the build hashes it like every source file)

Why generated: the loop is 128 MFMAs per K tile with exactly one other instruction between two MFMAs, on fixed registers
(accumulators a0..a255, fragment double buffer v0..v127) -- placing ~330 lines by hand once per schedule experiment is what
this script is for.  The SCHEDULE table below is the whole design; everything else is bookkeeping.

Structure (one wave per SIMD, 128 x 128 wave tile = 8 x 8 MFMA tiles of 16x16x32, K tile = 2 k-steps of 64 MFMAs):
  slot n = "in front of MFMA n" of the K tile.  k-step 0 (slots 0..63) computes on fragment buffer 0, k-step 1 on buffer 1.
  W = weight operand, A = activation operand; stage s holds tile t, tile t+1 is landing in stage s^1, tile t+2 is requested
  into stage s as soon as ALL waves have the corresponding operand of tile t in registers (one barrier per operand):
     1..15   ds_read W k-step-1 fragments of tile t           -> buffer 1
     20      lgkmcnt(0) + barrier: W(t) is in registers everywhere -> W region of stage s is free
     23..59  8 x LDS-DMA W(t+2) -> stage s, interleaved with ds_read A k-step-1 fragments (25..43)
     51      lgkmcnt(0) + barrier: A(t) is in registers everywhere -> A region of stage s is free
     62..125 8 x LDS-DMA A(t+2) -> stage s
     68      vmcnt(18) + barrier: W(t+1) has landed (own pieces by count, the others' by the barrier)
     70..84  ds_read W k-step-0 fragments of tile t+1          -> buffer 0 (its last reader was MFMA 63)
     105     vmcnt(15) + barrier: A(t+1) has landed
     107..121 ds_read A k-step-0 fragments of tile t+1         -> buffer 0
  No branch inside the tile: the requests of tiles >= nk go through a buffer resource with num_records = 0 (no traffic, the
  vmcnt bookkeeping stays uniform)."""
import os

# schedule experiments (the committed .inc is generated with the defaults): EA_W4A_TOP=prog -> progressive lgkmcnt waits in front of
# the first eight MFMAs of a tile instead of one full drain; EA_W4A_BAR1=<slot> -> the "W(t) consumed" barrier (default 20)
TOP_PROG = os.environ.get("EA_W4A_TOP", "drain") == "prog"
BAR1 = int(os.environ.get("EA_W4A_BAR1", "20"))

OUT = os.environ.get("EA_GEN_OUT") or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "easyanimate_amd", "csrc", "ea_gemm_w4_loop.inc")     # EA_GEN_OUT: write somewhere else (tests/test_w4a_isa_cpu.py compares with the committed file)

# fixed registers
FW = [0, 64]      # first VGPR of W fragment buffer b (8 fragments x 4 registers)
FA = [32, 96]     # first VGPR of A fragment buffer b
V_WK0, V_WK1, V_AK0, V_AK1 = 128, 129, 130, 131   # LDS byte addresses of the fragment reads (current stage), toggled by xor 0x8000
ADDR = {"wk0": V_WK0, "wk1": V_WK1, "ak0": V_AK0, "ak1": V_AK1}
S_RA, S_RW = 80, 84              # buffer resources (4 SGPRs each)
S_KA, S_KW = 88, 89              # scalar byte offset of the tile being requested
S_MW, S_MA = 90, 91              # LDS byte address of this wave's first W / A piece in the stage being filled
S_LEFT = 92                      # K tiles left, including the current one
S_NRA, S_NRW = 93, 94            # real num_records
S_KSTA, S_KSTW = 95, 100         # bytes to the next K tile
S_RB = 96                        # bias, then gate buffer resource (4 SGPRs; the _BG variant)
V_BIAS, V_GATE = 132, 164        # the _BG variant's outputs: the lane's 8 bias / 8 gate vectors (4 columns each), v132..v163 / v164..v195


def acc(i, j):
    """First AGPR of accumulator tile (row block i, column block j)."""
    return 4 * (j * 8 + i)


def mfma(b, n, swap=False, zero_c=False):
    """swap = False: the weight fragment is the MFMA's A operand (C^T orientation: a lane ends with 4 consecutive output columns of
    one row); swap = True: the activation fragment is (a lane ends with 4 consecutive ROWS of one column -- the V^T tiles of the
    fused QKV kernel)."""
    j, i = n >> 3, n & 7
    a = acc(i, j)
    w, x = f"v[{FW[b] + 4 * j}:{FW[b] + 4 * j + 3}]", f"v[{FA[b] + 4 * i}:{FA[b] + 4 * i + 3}]"
    c = "0" if zero_c else f"a[{a}:{a + 3}]"
    return f"v_mfma_f32_16x16x32_bf16 a[{a}:{a + 3}], {x}, {w}, {c}" if swap else f"v_mfma_f32_16x16x32_bf16 a[{a}:{a + 3}], {w}, {x}, {c}"


def rd(oper, b, x, addr):
    base = (FW if oper == "w" else FA)[b] + 4 * x
    off = f" offset:{x * 2048}" if x else ""
    return f"ds_read_b128 v[{base}:{base + 3}], v{ADDR[addr]}{off}"


def dma(oper, x):
    rs, k = (S_RW, S_KW) if oper == "w" else (S_RA, S_KA)
    return f"buffer_load_dwordx4 %[{oper}off{x}], s[{rs}:{rs + 3}], s{k} offen lds"


def m0_first(oper):
    return f"s_mov_b32 m0, s{S_MW if oper == 'w' else S_MA}"


M0_NEXT = "s_add_u32 m0, m0, 0x400"


def schedule():
    """slot -> instructions placed in front of MFMA `slot` of the K tile."""
    s = {n: [] for n in range(129)}
    for x in range(8):                                   # W k-step 1 of tile t
        s[1 + 2 * x].append(rd("w", 1, x, "wk1"))
    s[17].append(f"v_xor_b32 v{ADDR['wk1']}, 0x8000, v{ADDR['wk1']}")     # -> the other stage, for the next tile
    s[BAR1] += ["s_waitcnt lgkmcnt(0)", "s_barrier"]
    s[BAR1 + 1].append(m0_first("w"))
    w_slots = [BAR1 + 3, BAR1 + 6, BAR1 + 9, BAR1 + 12, BAR1 + 15, 53, 56, 59]
    for x, n in enumerate(w_slots):                       # W(t+2)
        s[n].append(dma("w", x))
        if x < 7:
            s[n + 1].append(M0_NEXT)
    a1_slots = [25, 28, 31, 34, 37, 39, 41, 43]
    for x, n in enumerate(a1_slots):                      # A k-step 1 of tile t
        s[n].append(rd("a", 1, x, "ak1"))
    s[45].append(f"v_xor_b32 v{ADDR['ak1']}, 0x8000, v{ADDR['ak1']}")
    s[47].append(f"v_xor_b32 v{ADDR['wk0']}, 0x8000, v{ADDR['wk0']}")
    s[49].append(f"v_xor_b32 v{ADDR['ak0']}, 0x8000, v{ADDR['ak0']}")
    s[51] += ["s_waitcnt lgkmcnt(0)", "s_barrier"]
    s[60].append(m0_first("a"))
    a_slots = [62, 65, 86, 88, 90, 97, 101, 125]
    for x, n in enumerate(a_slots):                       # A(t+2)
        s[n].append(dma("a", x))
        if x < 7:
            s[n + 1].append(M0_NEXT)
    s[68] += ["s_waitcnt vmcnt(18)", "s_barrier"]
    for x in range(8):                                    # W k-step 0 of tile t+1
        s[70 + 2 * x].append(rd("w", 0, x, "wk0"))
    s[105] += ["s_waitcnt vmcnt(15)", "s_barrier"]
    for x in range(8):                                    # A k-step 0 of tile t+1
        s[107 + 2 * x].append(rd("a", 0, x, "ak0"))
    # scalar bookkeeping of the next tile, behind the last request
    s[126] += [f"s_add_u32 s{S_KA}, s{S_KA}, s{S_KSTA}", f"s_add_u32 s{S_KW}, s{S_KW}, s{S_KSTW}"]
    s[127] += [f"s_xor_b32 s{S_MW}, s{S_MW}, 0x8000", f"s_xor_b32 s{S_MA}, s{S_MA}, 0x8000", f"s_sub_u32 s{S_LEFT}, s{S_LEFT}, 1"]
    return s


def check(s):
    """The counts the waits rely on."""
    order = []
    for n in range(129):
        for ins in s[n]:
            order.append((n, ins))
    dmas = [(n, i) for n, i in order if i.startswith("buffer_load")]
    assert len(dmas) == 16 and all("woff" in i for _, i in dmas[:8]) and all("aoff" in i for _, i in dmas[8:])
    before68 = sum(1 for n, _ in dmas if n < 68)
    before105 = sum(1 for n, _ in dmas if n < 105)
    assert before68 == 10 and before105 == 15, (before68, before105)     # vmcnt(8 + 10) / vmcnt(15)
    reads = [(n, i) for n, i in order if i.startswith("ds_read")]
    assert len(reads) == 32
    # a register written by a read must not be read by an MFMA that is still to come in the same role
    for n, i in reads:
        b = 1 if (f"v{V_WK1}" in i.split("],")[1] or f"v{V_AK1}" in i.split("],")[1]) else 0
        first_use, last_use_prev = (64, 128) if b == 1 else (128, 64)   # buffer 0 is next used by the NEXT tile's MFMA 0
        if b == 1:
            assert n < 64
        else:
            assert n >= 64
    # m0 is written at least one instruction before each DMA and never between the DMA and its own m0
    return True


def loop_body(s, swap, bias_gate=False):
    """The whole asm block (prologue, K loop, drain) as a list of instructions.

    bias_gate = True (EA_W4A_MAINLOOP_ASM_BG, gemm256_w4a_kernel): the lane's epilogue vectors -- bias and gate of its 4 columns in each of
    the 8 column blocks -- are fetched by 16 buffer loads in front of the first operand requests into FIXED registers that the asm
    statement declares as outputs ("={v[132:135]}" ..), so the epilogue never waits for a load: they are older than every operand
    request (the in-order vmcnt bookkeeping of the loop is untouched) and complete long before the loop ends.  A missing bias / gate
    is a resource with num_records = 0 (zeros, no traffic); columns past N read as zeros the same way."""
    body = []
    B = body.append
    # ---- prologue: resources, tile 0 and tile 1 requests, accumulators := 0, first fragments
    B(f"s_mov_b32 s{S_RA}, %[a_lo]")
    B(f"s_mov_b32 s{S_RA + 1}, %[a_hi]")
    B(f"s_mov_b32 s{S_RA + 2}, %[a_ext]")
    B(f"s_mov_b32 s{S_RA + 3}, 0x00020000")
    B(f"s_mov_b32 s{S_RW}, %[w_lo]")
    B(f"s_mov_b32 s{S_RW + 1}, %[w_hi]")
    B(f"s_mov_b32 s{S_RW + 2}, %[w_ext]")
    B(f"s_mov_b32 s{S_RW + 3}, 0x00020000")
    B(f"s_mov_b32 s{S_NRA}, %[a_ext]")
    B(f"s_mov_b32 s{S_NRW}, %[w_ext]")
    B(f"s_mov_b32 s{S_KSTA}, %[a_kst]")
    B(f"s_mov_b32 s{S_KSTW}, %[w_kst]")
    B(f"s_mov_b32 s{S_LEFT}, %[nk]")
    B(f"s_mov_b32 s{S_MW}, %[lds_w]")
    B(f"s_mov_b32 s{S_MA}, %[lds_a]")
    B(f"s_mov_b32 s{S_KA}, 0")
    B(f"s_mov_b32 s{S_KW}, 0")
    B(f"v_mov_b32 v{V_WK0}, %[wk0]")
    B(f"v_mov_b32 v{V_AK0}, %[ak0]")
    B(f"v_xor_b32 v{V_WK1}, 64, v{V_WK0}")        # k-step 1 = 16-byte chunk index + 4 under the XOR swizzle
    B(f"v_xor_b32 v{V_AK1}, 64, v{V_AK0}")

    def request_tile():
        for oper in ("w", "a"):
            B(m0_first(oper))
            for x in range(8):
                B("s_nop 0")
                B(dma(oper, x))
                if x < 7:
                    B(M0_NEXT)
    request_tile()                                     # tile 0 -> stage 0
    # tile 1 -> stage 1 (nk == 1: through an empty resource)
    B(f"s_cmp_gt_u32 s{S_LEFT}, 1")
    B(f"s_cselect_b32 s{S_RA + 2}, s{S_NRA}, 0")
    B(f"s_cselect_b32 s{S_RW + 2}, s{S_NRW}, 0")
    B(f"s_mov_b32 s{S_KA}, s{S_KSTA}")
    B(f"s_mov_b32 s{S_KW}, s{S_KSTW}")
    B(f"s_xor_b32 s{S_MW}, s{S_MW}, 0x8000")
    B(f"s_xor_b32 s{S_MA}, s{S_MA}, 0x8000")
    request_tile()
    B(f"s_xor_b32 s{S_MW}, s{S_MW}, 0x8000")          # the loop's first requests (tile 2) go to stage 0 again
    B(f"s_xor_b32 s{S_MA}, s{S_MA}, 0x8000")
    B(f"s_add_u32 s{S_KA}, s{S_KA}, s{S_KSTA}")
    B(f"s_add_u32 s{S_KW}, s{S_KW}, s{S_KSTW}")
    nbg = 0
    if bias_gate:                                      # behind the operand requests: tile 0 does not queue up behind them
        nbg = 16
        for what, v0 in (("b", V_BIAS), ("g", V_GATE)):
            B(f"s_mov_b32 s{S_RB}, %[{what}_lo]")
            B(f"s_mov_b32 s{S_RB + 1}, %[{what}_hi]")
            B(f"s_mov_b32 s{S_RB + 2}, %[{what}_ext]")
            B(f"s_mov_b32 s{S_RB + 3}, 0x00020000")
            for j in range(8):
                off = f" offset:{j * 64}" if j else ""
                B(f"buffer_load_dwordx4 v[{v0 + 4 * j}:{v0 + 4 * j + 3}], %[boff], s[{S_RB}:{S_RB + 3}], 0 offen{off}")
    B(f"s_waitcnt vmcnt({16 + nbg})")                  # tile 0 (this wave's pieces) has landed; tile 1 (and the bias / gate vectors) stay in flight
    B("s_barrier")
    for x in range(8):
        B(rd("w", 0, x, "wk0"))
    for x in range(8):
        B(rd("a", 0, x, "ak0"))

    def k_tile(first):
        """One K tile.  first: the peeled tile 0 -- its k-step-0 MFMAs take C = 0 (no accumulator initialisation anywhere), and its
        two vmcnt waits allow for the bias / gate loads, which are younger than tile 1's requests and older than tile 2's."""
        B(f"s_cmp_gt_u32 s{S_LEFT}, 2")                # tile t + 2 exists?
        B(f"s_cselect_b32 s{S_RA + 2}, s{S_NRA}, 0")
        B(f"s_cselect_b32 s{S_RW + 2}, s{S_NRW}, 0")
        if not TOP_PROG:
            B("s_waitcnt lgkmcnt(0)")                  # buffer 0 = k-step 0 of this tile
        for n in range(128):
            if TOP_PROG and n < 8:
                # MFMA n needs activation fragment n (the eight A reads are the youngest of the previous tile, in order); W k-step-1
                # reads issued at the odd slots before this point are younger still
                B(f"s_waitcnt lgkmcnt({7 - n + n // 2})")
            for ins in s[n]:
                if first and nbg and ins.startswith("s_waitcnt vmcnt("):
                    ins = f"s_waitcnt vmcnt({int(ins[len('s_waitcnt vmcnt('):-1]) + nbg})"
                B(ins)
            B(mfma(0 if n < 64 else 1, n & 63, swap, zero_c=first and n < 64))
        for ins in s[128]:
            B(ins)

    k_tile(True)
    B(f"s_cmp_lg_u32 s{S_LEFT}, 0")
    B("s_cbranch_scc0 9f")                             # nk == 1
    # ---- the K loop (tiles 1 ..)
    B("1:")
    k_tile(False)
    B(f"s_cmp_lg_u32 s{S_LEFT}, 0")
    B("s_cbranch_scc1 1b")
    B("9:")
    # ---- everything this wave requested has landed (the empty requests of the last two tiles too), the last MFMAs have left the pipe
    B("s_waitcnt vmcnt(0) lgkmcnt(0)")
    B("s_nop 15")
    B("s_nop 15")
    return body


def emit():
    s = schedule()
    check(s)
    L = []
    A = L.append
    A("// GENERATED by tools/gen_gemm_w4_asm.py -- do not edit; see that file for the schedule.")
    n_lines = 0
    for swap, bg in ((False, False), (True, False), (False, True)):
        A("#define EA_W4A_MAINLOOP_ASM_BG \\" if bg else "#define EA_W4A_MAINLOOP_ASM_SWAP \\" if swap else "#define EA_W4A_MAINLOOP_ASM \\")
        body = loop_body(s, swap, bg)
        n_lines = len(body)
        for ins in body:
            A(f'    "{ins}\\n\\t" \\')
        L[-1] = L[-1][:-2]      # no continuation behind the last line
        A("")
    # The 256 accumulators are OUTPUT operands of the main asm (EA_W4A_ACC_OUTPUTS: f32x4_t accq[64] bound to a[4n:4n+3]) and INPUT
    # operands of the read-out statements -- the register allocator knows they are live in between (as clobbers only,
    # nothing but tests/test_w4a_isa_cpu.py stood between a compiler spill into an AGPR and silently wrong results)
    clob = [f'"v{r}"' for r in range(132)] + [f'"s{r}"' for r in range(80, 102)] + ['"m0"', '"scc"', '"memory"']
    A("#define EA_W4A_ACC_OUTPUTS(accq) \\")
    outs = [f'"={{a[{4 * n}:{4 * n + 3}]}}"(accq[{n}])' for n in range(64)]
    for k in range(0, 64, 4):
        A("    " + ", ".join(outs[k:k + 4]) + (", \\" if k + 4 < 64 else ""))
    A("")
    A("#define EA_W4A_CLOBBERS \\")
    for k in range(0, len(clob), 16):
        A("    " + ", ".join(clob[k:k + 16]) + (", \\" if k + 16 < len(clob) else ""))
    A("")
    # ---- the _BG variant's output operands: f32x4_t bias[8], gate[8]
    A("#define EA_W4A_BG_OUTPUTS(bias, gate) \\")
    outs = [f'"=&{{v[{V_BIAS + 4 * j}:{V_BIAS + 4 * j + 3}]}}"(bias[{j}])' for j in range(8)] + [f'"=&{{v[{V_GATE + 4 * j}:{V_GATE + 4 * j + 3}]}}"(gate[{j}])' for j in range(8)]
    for k in range(0, 16, 4):
        A("    " + ", ".join(outs[k:k + 4]) + (", \\" if k + 4 < 16 else ""))
    A("")
    # ---- accumulator read-out: column half h (j = 4h .. 4h + 3) into f32x4_t dst[8][4]
    for h in range(2):
        A(f"#define EA_W4A_READ_HALF{h}(dst, accq) \\")
        lines = []
        for jj in range(4):
            for i in range(8):
                a = acc(i, 4 * h + jj)
                for e in range(4):
                    assert a % 4 == 0
                    if e == 0:
                        lines.append(f'    dst[{i}][{jj}] = accq[{a // 4}];')
        for k, ln in enumerate(lines):
            A(ln + (" \\" if k + 1 < len(lines) else ""))
        A("")
    open(OUT, "w").write("\n".join(L))
    print(OUT, n_lines, "asm lines per variant")


if __name__ == "__main__":
    emit()
