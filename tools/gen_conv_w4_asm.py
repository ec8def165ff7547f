"""Generates easyanimate_amd/csrc/ea_conv_w4_loop.inc: the hand-placed main loop of conv3d_cl_row16_w4a_kernel (ea_conv.hip) -- the
3x3x3 row-slab implicit GEMM over 32-channel stages on FOUR waves, one per SIMD, 128 voxels x 128 channels per wave (8 x 8 MFMA
tiles of 16x16x32 = the whole AGPR file), as ONE inline-asm block per tile geometry.

    python tools/gen_conv_w4_asm.py

The same idea as tools/gen_gemm_w4_asm.py (exactly one non-MFMA instruction between two MFMAs, no branch inside the loop body,
requests beyond the end go through empty buffer resources), on the row-slab data flow of conv3d_cl_row16_k32_kernel:

  * a SLAB = one input row segment (TM + 2 voxels x 32 channels, 64-byte LDS rows) for one (dt, dh, channel block); its three
    dw taps are three shifted fragment reads of the same staged rows.  A TILE = (slab, dw) = one k32 step = 64 MFMAs per wave.
  * two A stages (slab s in stage s & 1), three W stages (the stage of a tile is its dw).
  * the fragments of tile t + 1 are read (into the other half of a register double buffer) while tile t is computed; ONE barrier
    per tile, a few MFMAs into it: behind it (a) everything tile t + 1 needs has landed on every wave (own pieces by vmcnt, the
    others' by the barrier) and (b) every wave holds tile t's fragments in registers, so tile t's W stage is free for the W tile
    of (slab + 1, dw), and -- at dw = 2 -- the slab's A stage is free for slab + 2.
  * 3 tiles per slab is odd, so the register double buffer changes parity per slab: the loop body is TWO slabs (384 MFMAs).
  * the (dt, dh) -> input row table (base address, extent; extent 0 = a row of zero padding) lives in three VGPRs, lane = dt*3+dh,
    and is read with v_readlane when the request cursor moves to the next (dt, dh).

Fixed registers: accumulators a0..a255 (acc(i, j) = 4 * (8 j + i), the read-out macros of ea_gemm_w4_loop.inc apply), fragment
double buffer v0..v127, fragment addresses v128..v131 (a_k[dw = 0, 1, 2], w_k), scalars s80..s101."""
import os

OUT = os.environ.get("EA_GEN_OUT") or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "easyanimate_amd", "csrc", "ea_conv_w4_loop.inc")     # EA_GEN_OUT: write somewhere else (tests/test_w4a_isa_cpu.py compares with the committed file)

FW = [0, 64]
FA = [32, 96]
V_AK = [128, 129, 130]
V_WK = 131
S_RA, S_RW = 80, 84
S_ASOFF = 88      # scalar byte offset (channel block) of the slab being requested (slab s + 2)
S_KOFF1 = 89      # W scalar byte offset of (slab s + 1, dw = 0)
S_MW, S_MA = 90, 91
S_LEFT = 92       # slabs left, including the current one
S_WEXT = 93
S_CBLK = 94
S_CIN2 = 95       # C_in * 2: bytes between two taps of a weight row
S_D1, S_C1 = 96, 97      # cursor of slab s + 1: (dt*3+dh, channel block)
S_D2, S_C2 = 98, 99      # cursor of slab s + 2
S_T0, S_T1 = 100, 101    # temporaries


def acc(i, j):
    return 4 * (j * 8 + i)


def mfma(b, n):
    j, i = n >> 3, n & 7
    a = acc(i, j)
    return f"v_mfma_f32_16x16x32_bf16 a[{a}:{a + 3}], v[{FW[b] + 4 * j}:{FW[b] + 4 * j + 3}], v[{FA[b] + 4 * i}:{FA[b] + 4 * i + 3}], a[{a}:{a + 3}]"


class Geo:
    def __init__(self, name, TM, BN, cb=False):
        self.name, self.TM, self.BN = name, TM, BN
        # cb (round 6): the input is CHANNEL-BLOCKED, [C_in / 32][T][H][W][32] -- a slab's 16-voxel piece is 1 KiB of consecutive
        # memory instead of sixteen 64-byte runs a voxel (C_in * 2 bytes) apart.  The table rows then describe channel block 0 and the
        # slab's block moves the resource BASE (64-bit: a block is T * H * W * 64 bytes, beyond any 32-bit offset); the scalar offset
        # of the request stays 0.
        self.cb = cb
        self.NROW = TM + 2
        self.NPIECE = (self.NROW + 15) // 16
        self.PPW = (self.NPIECE + 3) // 4            # A pieces (1 KiB = 16 rows) per wave and slab
        self.WP = BN // 64                           # W pieces per wave and tile
        self.W_BYTES = BN * 64
        self.A_XOR = 0x10000 if TM == 512 else 0x8000
        self.vm = [self.WP + self.PPW, self.WP + self.PPW, self.WP]


def rd_w(b, j, dw_next, g):
    base = FW[b] + 4 * j
    off = dw_next * g.W_BYTES + j * 1024
    return f"ds_read_b128 v[{base}:{base + 3}], v{V_WK}" + (f" offset:{off}" if off else "")


def rd_a(b, i, dw_next):
    base = FA[b] + 4 * i
    return f"ds_read_b128 v[{base}:{base + 3}], v{V_AK[dw_next]}" + (f" offset:{i * 1024}" if i else "")


def dma_w(i):
    return f"buffer_load_dwordx4 %[woff{i}], s[{S_RW}:{S_RW + 3}], s{S_T0} offen lds"


def dma_a(i):
    return f"buffer_load_dwordx4 %[aoff{i}], s[{S_RA}:{S_RA + 3}], s{S_ASOFF} offen lds"


def advance(d, c):
    """cursor (dt*3+dh, channel block) -> the next slab."""
    return [f"s_add_u32 s{c}, s{c}, 1", f"s_cmp_eq_u32 s{c}, s{S_CBLK}", f"s_cselect_b32 s{c}, 0, s{c}", f"s_cselect_b32 s{S_T1}, 1, 0",
            f"s_add_u32 s{d}, s{d}, s{S_T1}"]


def load_row(d):
    """A resource words 0..2 of table entry s<d> (the extent is overridden by the caller when the slab does not exist)."""
    return [f"v_readlane_b32 s{S_RA}, %[t_lo], s{d}", f"v_readlane_b32 s{S_RA + 1}, %[t_hi], s{d}", f"v_readlane_b32 s{S_RA + 2}, %[t_ext], s{d}"]


def block_base(c):
    """channel-blocked input: resource base += (channel block s<c>) * (bytes per block, %[cb_lo] / %[cb_hi]), 64-bit."""
    return [f"s_mul_hi_u32 s{S_T1}, s{c}, %[cb_lo]", f"s_mul_i32 s{S_T0}, s{c}, %[cb_lo]", f"s_add_u32 s{S_RA}, s{S_RA}, s{S_T0}",
            f"s_addc_u32 s{S_RA + 1}, s{S_RA + 1}, s{S_T1}", f"s_mul_i32 s{S_T0}, s{c}, %[cb_hi]", f"s_add_u32 s{S_RA + 1}, s{S_RA + 1}, s{S_T0}"]


def asoff(g, c):
    """scalar byte offset of a slab's request: the channel block inside a voxel's channels -- or nothing (blocked input)."""
    return f"s_mov_b32 s{S_ASOFF}, 0" if g.cb else f"s_lshl_b32 s{S_ASOFF}, s{c}, 6"


def tile(g, dw, par):
    """slot -> instructions in front of MFMA `slot` of tile (slab, dw), computing on fragment buffer `par`."""
    s = {n: [] for n in range(65)}
    nb = par ^ 1
    dwn = (dw + 1) % 3
    s[0].append("s_waitcnt lgkmcnt(0)")
    s[4] += [f"s_waitcnt vmcnt({g.vm[dw]})", "s_barrier"]
    # fragments of the next tile: activations first (all eight feed the first eight MFMAs), then weights; one per two slots
    for i in range(8):
        s[6 + 2 * i].append(rd_a(nb, i, dwn))
    for j in range(8):
        s[22 + 2 * j].append(rd_w(nb, j, dwn, g))
    s[23].append(f"v_xor_b32 v{V_AK[dwn]}, 0x{g.A_XOR:x}, v{V_AK[dwn]}")      # the reads above were this slab's (or, dw = 2, the next one's) last of that tap
    # requests: the W tile of (slab + 1, dw) into this tile's W stage; at dw = 2 the A slab s + 2 into this slab's stage
    reqs = []
    reqs.append([f"s_add_u32 s{S_T0}, s{S_KOFF1}, s{S_T1}"] if dw else [f"s_mov_b32 s{S_T0}, s{S_KOFF1}"])
    for i in range(g.WP):
        reqs.append(("m0", f"s_add_u32 m0, s{S_MW}, 0x{dw * g.W_BYTES + i * 1024:x}"))
        reqs.append(("dma", dma_w(i)))
    if dw == 2:
        for i in range(g.PPW):
            reqs.append(("m0", f"s_add_u32 m0, s{S_MA}, 0x{i * 1024:x}"))
            reqs.append(("dma", dma_a(i)))
    slot = 5
    if dw == 1:
        s[3].append(f"s_mov_b32 s{S_T1}, s{S_CIN2}")
    if dw == 2:
        s[3].append(f"s_lshl_b32 s{S_T1}, s{S_CIN2}, 1")
    for r in reqs:                      # one instruction per slot: an LDS-DMA request sits one MFMA behind its m0 write
        if isinstance(r, list):
            s[slot] += r
        else:
            s[slot].append(r[1])
        slot += 1
    assert slot <= 38, slot
    return s


def slab_tail(g):
    """Scalar bookkeeping behind the last request of a slab (in front of the last MFMAs of tile dw = 2): cursors, validity of the
    next requests, the A resource of slab s + 3 -> s + 2 of the next slab."""
    t = []
    t += advance(S_D1, S_C1)
    t += advance(S_D2, S_C2)
    t += [f"s_sub_u32 s{S_LEFT}, s{S_LEFT}, 1", f"s_xor_b32 s{S_MA}, s{S_MA}, 0x{g.A_XOR:x}"]
    # W offset of the new slab s + 1: (d1 * 3 * C_in + c1 * 32) * 2 bytes
    t += [f"s_mul_i32 s{S_T0}, s{S_D1}, 3", f"s_mul_i32 s{S_T0}, s{S_T0}, s{S_CIN2}", f"s_lshl_b32 s{S_T1}, s{S_C1}, 6", f"s_add_u32 s{S_KOFF1}, s{S_T0}, s{S_T1}",
          asoff(g, S_C2)]
    return t


def validity(g):
    """Resource extents for the coming slab's requests: W needs slab s + 1 (left > 1), A needs slab s + 2 (left > 2)."""
    return [f"s_cmp_gt_u32 s{S_LEFT}, 1", f"s_cselect_b32 s{S_RW + 2}, s{S_WEXT}, 0"] + load_row(S_D2) + \
           ["s_nop 1", f"s_cmp_gt_u32 s{S_LEFT}, 2", f"s_cselect_b32 s{S_RA + 2}, s{S_RA + 2}, 0"] + (block_base(S_C2) if g.cb else [])


def body(g):
    B = []
    A = B.append
    # ---- prologue
    A(f"s_mov_b32 s{S_RW}, %[w_lo]")
    A(f"s_mov_b32 s{S_RW + 1}, %[w_hi]")
    A(f"s_mov_b32 s{S_RW + 2}, %[w_ext]")
    A(f"s_mov_b32 s{S_RW + 3}, 0x00020000")
    A(f"s_mov_b32 s{S_RA + 3}, 0x00020000")
    A(f"s_mov_b32 s{S_WEXT}, %[w_ext]")
    A(f"s_mov_b32 s{S_CBLK}, %[cblocks]")
    A(f"s_mov_b32 s{S_CIN2}, %[cin2]")
    A(f"s_mov_b32 s{S_LEFT}, %[nslabs]")
    A(f"s_mov_b32 s{S_MW}, %[lds_w]")
    A(f"s_mov_b32 s{S_MA}, %[lds_a]")
    A(f"v_mov_b32 v{V_AK[0]}, %[ak0]")
    A(f"v_mov_b32 v{V_AK[1]}, %[ak1]")
    A(f"v_mov_b32 v{V_AK[2]}, %[ak2]")
    A(f"v_mov_b32 v{V_WK}, %[wk]")
    # slab 0 -> A stage 0
    A(f"s_mov_b32 s{S_D2}, 0")
    A(f"s_mov_b32 s{S_C2}, 0")
    A(f"s_mov_b32 s{S_ASOFF}, 0")
    for ins in load_row(S_D2):
        A(ins)
    A("s_nop 4")                                    # VALU-written SGPRs -> VMEM
    for i in range(g.PPW):
        A(f"s_add_u32 m0, s{S_MA}, 0x{i * 1024:x}")
        A("s_nop 0")
        A(dma_a(i))
    # the three W tiles of slab 0 -> W stages 0, 1, 2
    for dw in range(3):
        A(f"s_mul_i32 s{S_T0}, s{S_CIN2}, {dw}")
        for i in range(g.WP):
            A(f"s_add_u32 m0, s{S_MW}, 0x{dw * g.W_BYTES + i * 1024:x}")
            A("s_nop 0")
            A(dma_w(i))
    # slab 1 -> A stage 1
    for ins in advance(S_D2, S_C2):
        A(ins)
    A(asoff(g, S_C2))
    A("s_nop 3")
    for ins in load_row(S_D2):
        A(ins)
    A(f"s_xor_b32 s{S_MA}, s{S_MA}, 0x{g.A_XOR:x}")
    A("s_nop 4")
    if g.cb:
        for ins in block_base(S_C2):
            A(ins)
        A("s_nop 0")
    for i in range(g.PPW):
        A(f"s_add_u32 m0, s{S_MA}, 0x{i * 1024:x}")
        A("s_nop 0")
        A(dma_a(i))
    A(f"s_xor_b32 s{S_MA}, s{S_MA}, 0x{g.A_XOR:x}")           # back to slab 0's stage: the first in-loop A request (slab 2) goes there
    # cursors: slab 1 (W requests of the first slab), slab 2 (its A request)
    A(f"s_mov_b32 s{S_D1}, s{S_D2}")
    A(f"s_mov_b32 s{S_C1}, s{S_C2}")
    for ins in advance(S_D2, S_C2):
        A(ins)
    A(f"s_mul_i32 s{S_T0}, s{S_D1}, 3")
    A(f"s_mul_i32 s{S_T0}, s{S_T0}, s{S_CIN2}")
    A(f"s_lshl_b32 s{S_T1}, s{S_C1}, 6")
    A(f"s_add_u32 s{S_KOFF1}, s{S_T0}, s{S_T1}")
    A(asoff(g, S_C2))
    A("s_nop 3")
    for ins in validity(g):
        A(ins)
    for r in range(256):
        A(f"v_accvgpr_write_b32 a{r}, 0")
    A("s_waitcnt vmcnt(0)")
    A("s_barrier")
    for i in range(8):                                          # fragments of tile (0, 0)
        A(rd_a(0, i, 0))
    for j in range(8):
        A(rd_w(0, j, 0, g))
    A(f"v_xor_b32 v{V_AK[0]}, 0x{g.A_XOR:x}, v{V_AK[0]}")
    # ---- the loop: two slabs per trip
    A("1:")
    t = 0
    for slab in range(2):
        for dw in range(3):
            sch = tile(g, dw, t & 1)
            last = slab_tail(g) + validity(g) if dw == 2 else []
            # the scalar tail goes one instruction per slot behind the tile's last request (blocked input: six more scalar
            # instructions than slots -- the first ones go two to a slot)
            slot = 38                                           # behind the last fragment read (slot 36) and the last request
            extra = max(0, slot + len(last) - 64)
            assert extra <= 12, (slot, len(last))
            k = 0
            for n_ in range(slot, 64):
                take = 2 if n_ - slot < extra else 1
                sch[n_] += last[k:k + take]
                k += take
            assert k >= len(last)
            for n in range(64):
                for ins in sch[n]:
                    A(ins)
                A(mfma(t & 1, n))
            for ins in sch[64]:
                A(ins)
            t += 1
    A(f"s_cmp_lg_u32 s{S_LEFT}, 0")
    A("s_cbranch_scc1 1b")
    A("s_waitcnt vmcnt(0) lgkmcnt(0)")
    A("s_nop 15")
    A("s_nop 15")
    return B


def emit():
    L = ["// GENERATED by tools/gen_conv_w4_asm.py -- do not edit; see that file for the schedule."]
    for g in (Geo("M512", 512, 128), Geo("N256", 256, 256), Geo("M512_CB", 512, 128, cb=True), Geo("N256_CB", 256, 256, cb=True)):
        b = body(g)
        n_m = sum("v_mfma" in x for x in b)
        assert n_m == 384, n_m
        L.append(f"#define EA_CONV_W4A_ASM_{g.name} \\")
        for ins in b:
            L.append(f'    "{ins}\\n\\t" \\')
        L[-1] = L[-1][:-2]
        L.append("")
        L.append(f"#define EA_CONV_W4A_PPW_{g.name} {g.PPW}")
        L.append(f"#define EA_CONV_W4A_WP_{g.name} {g.WP}")
        L.append(f"#define EA_CONV_W4A_AXOR_{g.name} 0x{g.A_XOR:x}")
        L.append("")
        print(g.name, len(b), "asm lines; PPW", g.PPW, "WP", g.WP, "vmcnt", g.vm)
    # (the accumulators a0..a255 are output operands: EA_W4A_ACC_OUTPUTS of ea_gemm_w4_loop.inc)
    clob = [f'"v{r}"' for r in range(132)] + [f'"s{r}"' for r in range(80, 102)] + ['"m0"', '"scc"', '"memory"']
    L.append("#define EA_CONV_W4A_CLOBBERS \\")
    for k in range(0, len(clob), 16):
        L.append("    " + ", ".join(clob[k:k + 16]) + (", \\" if k + 16 < len(clob) else ""))
    L.append("")
    open(OUT, "w").write("\n".join(L))


if __name__ == "__main__":
    emit()
