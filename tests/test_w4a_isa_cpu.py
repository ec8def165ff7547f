"""Static check of the four-wave hand-placed GEMM / convolution kernels' ISA (no GPU needed: hipcc cross-compiles).  The main loop
is one inline-asm block on FIXED registers (accumulators a0..a255, fragments v0..v131).  Since round 6 the 256 accumulators are
OUTPUT operands of that asm (EA_W4A_ACC_OUTPUTS: f32x4 accq[64] bound to a[4n:4n+3]) and the epilogues read them as ordinary
values, so the register allocator knows they are live (before, they were clobbers only and this test was the one guard).
The test stays as an independent check of the compiled code:
  * no scratch memory, no branch inside the K loop, exactly 128 (384) MFMAs per loop body;
  * behind the main asm, the FIRST thing that happens to accumulator a[r] on every path is a read (its read-out); the compiler
    writes a[r] -- as spill space -- only after that, and reads back only what it wrote itself."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "easyanimate_amd", "csrc")
# (source file, kernel symbol fragment, MFMAs inside each hand-placed loop body)
KERNELS = [("ea_gemm.hip", "gemm256_w4a_kernelILi0E", [128]), ("ea_gemm.hip", "gemm256_w4a_kernelILi1E", [128]),
           ("ea_gemm.hip", "gemm256_w4a_kernelILi2E", [128]), ("ea_gemm.hip", "gemm256_qkv_w4a_kernel", [128, 128]),
           ("ea_conv.hip", "conv3d_cl_row16_w4a_kernelILi128ELi512ELb0E", [384]), ("ea_conv.hip", "conv3d_cl_row16_w4a_kernelILi256ELi256ELb0E", [384]),
           # the channel-blocked-input variants (round 6)
           ("ea_conv.hip", "conv3d_cl_row16_w4a_kernelILi128ELi512ELb1E", [384]), ("ea_conv.hip", "conv3d_cl_row16_w4a_kernelILi256ELi256ELb1E", [384])]


HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.fixture(scope="module")
def isa(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip(f"{HIPCC} not found: the ISA check needs the cross-compiler")
    d = tmp_path_factory.mktemp("isa")
    out = {}
    for src in sorted({k[0] for k in KERNELS}):
        o = str(d / (src + ".s"))
        r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-Wno-inline-asm",
                            "-x", "hip", "-S", "--cuda-device-only", os.path.join(CSRC, src), "-o", o], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        out[src] = open(o).read().split("\n")
    return out


def _regs(text):
    """AGPR indices named in an instruction's operands."""
    out = []
    for m in re.finditer(r"(?<![\w.$])a(\d+)\b", text):
        out.append(int(m.group(1)))
    for m in re.finditer(r"(?<![\w.$])a\[(\d+):(\d+)\]", text):
        out += list(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


ALL = (1 << 256) - 1


def _analyse(name, body):
    """Forward dataflow over the kernel's basic blocks.  Per AGPR two facts: `acc` = it MAY still hold an accumulator that has not
    been read out (union over the paths), `own` = it holds a value the compiler wrote itself on EVERY path.  The main asm makes all
    256 `acc`; a read-out asm clears its register's bit; a compiler write needs acc = 0, a compiler read needs own = 1."""
    # ---- events per line: ("main",), ("readout", r), ("write", r), ("read", r); control flow: labels and branches outside inline asm
    events, labels, succ_of_line = {}, {}, {}
    inasm, block_lines = False, []
    n_loop_mfma, loops_ok = [], True
    k = 0
    while k < len(body):
        t = body[k].strip()
        if t.startswith(";;#ASMSTART"):
            j = k
            while not body[j].strip().startswith(";;#ASMEND"):
                j += 1
            chunk = body[k + 1:j]
            if len(chunk) > 300:
                events[k] = [("main",)]
                first = [i for i, l in enumerate(chunk) if l.strip() == "1:"][0]
                back = [i for i, l in enumerate(chunk) if "s_cbranch_scc1 1b" in l][0]
                inner = chunk[first:back]
                n_loop_mfma.append(sum("v_mfma" in l for l in inner))
                loops_ok &= not any("s_cbranch" in l or "s_branch" in l for l in inner)
            else:
                ev = []
                for l in chunk:
                    m = re.search(r"v_accvgpr_read_b32 v\d+, a(\d+)", l)
                    if m:
                        ev.append(("readout", int(m.group(1))))
                    else:
                        assert not _regs(l.split(";")[0]) or "v_mbcnt" in l, l
                events[k] = ev
            k = j + 1
            continue
        code = body[k].split(";")[0].strip()
        m = re.match(r"(\.LBB\d+_\d+):", code)
        if m:
            labels[m.group(1)] = k
        elif code.startswith(("s_cbranch", "s_branch")):
            succ_of_line[k] = (code.split()[0], code.split()[1])
        elif code.startswith("s_endpgm"):
            succ_of_line[k] = ("end", None)
        elif "v_accvgpr_write" in code:
            events[k] = [("write", r) for r in _regs(code.split(",")[0])]
        elif _regs(code):
            events[k] = [("read", r) for r in _regs(code)]
        k += 1
    # ---- basic blocks: leaders = first line, label lines, lines behind a branch
    leaders = sorted({0} | set(labels.values()) | {k + 1 for k in succ_of_line if k + 1 < len(body)})
    blocks = []
    for a, b in zip(leaders, leaders[1:] + [len(body)]):
        succ = []
        last = max([k for k in succ_of_line if a <= k < b], default=None)
        if last is not None and last == max(k for k in range(a, b) if body[k].split(";")[0].strip()):
            kind, tgt = succ_of_line[last]
            if kind != "end":
                succ.append(labels[tgt])
                if kind != "s_branch":
                    succ.append(b)
        elif b < len(body):
            succ.append(b)
        blocks.append((a, b, succ))
    start_of = {a: i for i, (a, b, _) in enumerate(blocks)}
    state_in = {0: (0, 0)}
    work = [0]
    errors, n_w, n_r = [], 0, 0
    seen_checks = set()
    while work:
        bi = work.pop()
        a, b, succ = blocks[bi]
        acc, own = state_in[bi]
        for k in range(a, b):
            for ev in events.get(k, ()):
                if ev[0] == "main":
                    acc, own = ALL, 0
                elif ev[0] == "readout":
                    acc &= ~(1 << ev[1])
                elif ev[0] == "write":
                    if (acc >> ev[1]) & 1 and (k, ev) not in seen_checks:
                        errors.append(f"line {k}: compiler write to a{ev[1]}, which may hold an accumulator that was not read out: {body[k].strip()}")
                    seen_checks.add((k, ev))
                    own |= 1 << ev[1]
                elif ev[0] == "read" and (acc >> ev[1]) & 1:
                    acc &= ~(1 << ev[1])           # the compiler's read-out of an accumulator the main asm returned
                elif ev[0] == "read":
                    if not (own >> ev[1]) & 1 and (k, ev) not in seen_checks:
                        errors.append(f"line {k}: compiler read of a{ev[1]}, which it did not write on every path: {body[k].strip()}")
                    seen_checks.add((k, ev))
        for s_ in succ:
            if s_ not in start_of:
                continue
            si = start_of[s_]
            if si not in state_in:
                state_in[si] = (acc, own)
                work.append(si)
            else:
                oa, oo = state_in[si]
                na, no = oa | acc, oo & own
                if (na, no) != (oa, oo):
                    state_in[si] = (na, no)
                    work.append(si)
    n_w = sum(1 for evs in events.values() for e in evs if e[0] == "write")
    n_r = sum(1 for evs in events.values() for e in evs if e[0] == "read")
    return errors, n_w, n_r, n_loop_mfma, loops_ok


@pytest.mark.parametrize("src,name,loop_mfmas", KERNELS)
def test_compiler_leaves_the_accumulators_alone(isa, src, name, loop_mfmas):
    text = isa[src]
    st = [i for i, l in enumerate(text) if re.match(rf"_ZN\w*{name}\w*:", l)][0]
    en = [i for i in range(st, len(text)) if ".end_amdhsa_kernel" in text[i]][0]
    body = text[st:en]
    meta = {k: int([l.split()[-1] for l in body if k in l][0]) for k in (".amdhsa_private_segment_fixed_size", ".amdhsa_accum_offset")}
    assert meta[".amdhsa_private_segment_fixed_size"] == 0 and not any("scratch_" in l for l in body)
    if "gemm256_w4a_kernel" in name:
        # the bias / gate vectors land in v132..v195 (outputs of the main asm, written in its prologue): no input operand of the
        # asm -- the request offsets, the fragment addresses -- may have been placed there or in the clobbered v0..v131
        main = [l for l in body if re.search(r"buffer_load_dwordx4 v\d+, s\[\d+:\d+\], s\d+ offen lds", l)]
        assert len(main) >= 48
        assert all(int(re.search(r"buffer_load_dwordx4 v(\d+),", l).group(1)) >= 196 for l in main), main[:3]
        outs = [l for l in body if re.search(r"buffer_load_dwordx4 v\[\d+:\d+\], v\d+, s\[96:99\], 0 offen", l)]
        assert len(outs) == 16 and all(int(re.search(r", v(\d+), s\[96", l).group(1)) >= 196 for l in outs), outs[:3]
    errors, n_w, n_r, n_loop_mfma, loops_ok = _analyse(name, body)
    assert n_loop_mfma == loop_mfmas and loops_ok
    print(f"[isa] {name}: arch VGPRs {meta['.amdhsa_accum_offset']}, compiler AGPR spill writes / reads behind the main loop: {n_w} / {n_r}")
    assert not errors, "\n".join(errors[:10])


@pytest.mark.parametrize("gen,inc", [("gen_gemm_w4_asm.py", "ea_gemm_w4_loop.inc"), ("gen_conv_w4_asm.py", "ea_conv_w4_loop.inc")])
def test_committed_loops_are_what_the_generators_write(gen, inc, tmp_path):
    """The .inc files are generated (tools/gen_*_w4_asm.py, default switches) and committed: a hand edit, or a generator change
    without regenerating, would leave the build on code that the schedule tables no longer describe."""
    import sys
    out = str(tmp_path / inc)
    env = {k: v for k, v in os.environ.items() if not k.startswith("EA_W4A_")}
    env["EA_GEN_OUT"] = out
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", gen)], env=env, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-1500:]
    assert open(out).read() == open(os.path.join(CSRC, inc)).read()
