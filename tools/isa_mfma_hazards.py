"""Distance check for kernels whose MFMAs are inline asm (hipcc inserts no wait states for their results): in the kernel's
largest loop, for every MFMA that writes arch VGPRs, the number of instructions / MFMAs up to the first later instruction
that READS one of those registers (VALU, memory, another MFMA as A / B), and for every VALU write of a register that a
later MFMA reads, the distance to that MFMA; and the reverse (MFMA reads a register, a later VALU overwrites it).
    python tools/isa_mfma_hazards.py /tmp/att_v4.s attention_fwd_v4_kernel [first_label last_label]
Walks the basic blocks that contain MFMAs in layout order as one cyclic instruction stream (the tile loop)."""
import re
import sys


def regs(tok):
    m = re.fullmatch(r'([va])\[(\d+):(\d+)\]', tok)
    if m:
        return {(m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1)}
    m = re.fullmatch(r'([va])(\d+)', tok)
    return {(m.group(1), int(m.group(2)))} if m else set()


def main(path, kernel, first=None, last=None):
    s = open(path).read()
    m = re.search(r'\n(_Z\S*' + kernel + r'\S*?):[^\n]*\n(.*?)\.Lfunc_end', s, re.S)
    blocks = re.split(r'\n(?=\.LBB\d+_\d+:)', m.group(2))
    if first:     # explicit range of labels, e.g. .LBB7_11 .LBB7_10 (layout order; the loop may end in front of its head)
        names = [b.split(':')[0] for b in blocks]
        a, z = names.index(first), names.index(last)
        loop = blocks[a:z + 1] if a <= z else blocks[a:] + blocks[:z + 1]
        loop = [b for b in loop if 'v_mfma' in b or 'ds_read' in b or 'buffer_load' in b]
    else:
        loop = [b for b in blocks if len(re.findall(r'v_mfma', b)) >= 8][1:]     # skip the prologue block
        loop = [b for b in loop if 'v_exp_f32' in b or 'ds_read' in b]
    ins = []
    for b in loop:
        for l in b.split('\n'):
            l = re.sub(r'\s*;.*', '', l.strip())
            if not l or l.endswith(':') or l.startswith('.'):
                continue
            op, *rest = l.split(None, 1)
            ops = [t.strip() for t in rest[0].split(',')] if rest else []
            ins.append((op, ops))
    n = len(ins)
    print(f"{len(loop)} blocks, {n} instructions, {sum(1 for o, _ in ins if o.startswith('v_mfma'))} MFMAs in the loop")
    worst = {}

    def note(kind, d_ins, d_mfma, i, j):
        k = worst.get(kind)
        if k is None or d_ins < k[0]:
            worst[kind] = (d_ins, d_mfma, ins[i], ins[j])

    def parts(op, ops):
        writes = regs(ops[0]) if ops and not op.startswith(('ds_write', 'global_store', 'buffer_store', 's_')) else set()
        if op.startswith(('ds_write', 'global_store', 'buffer_store')):
            reads = set().union(*[regs(t) for t in ops]) if ops else set()
        else:
            reads = set().union(*[regs(t) for t in ops[1:]]) if len(ops) > 1 else set()
        return writes, reads

    for i, (op, ops) in enumerate(ins):
        if not ops:
            continue
        dst, _ = parts(op, ops)
        is_mfma = op.startswith('v_mfma')
        # (1) result of an MFMA in arch VGPRs -> its first reader that is not the accumulate of the same tile
        if is_mfma and dst and ops[0].startswith('v'):
            nm = 0
            for k in range(1, n):
                op2, ops2 = ins[(i + k) % n]
                w2, r2 = parts(op2, ops2)
                m2 = op2.startswith('v_mfma')
                same_acc = m2 and ops2[0] == ops[0] and ops2[3:4] == [ops[0]]
                if (dst & r2) and not same_acc:
                    note("MFMA result -> first reader (%s)" % ("MFMA A/B" if m2 else op2), k, nm, i, (i + k) % n)
                    break
                if (dst & w2) and not same_acc:
                    break
                nm += m2
        # (2) A / B operands of an MFMA -> the next instruction that overwrites one of them
        if is_mfma:
            ab = set().union(*[regs(t) for t in ops[1:3]])
            nm = 0
            for k in range(1, n):
                op2, ops2 = ins[(i + k) % n]
                w2, _ = parts(op2, ops2)
                if ab & w2:
                    note("MFMA reads A/B -> overwritten by " + op2, k, nm, i, (i + k) % n)
                    break
                nm += op2.startswith('v_mfma')
        # (3) VALU / LDS write -> the first MFMA that reads it
        if not is_mfma and dst and op.startswith(('v_', 'ds_read')):
            nm = 0
            for k in range(1, n):
                op2, ops2 = ins[(i + k) % n]
                w2, r2 = parts(op2, ops2)
                if op2.startswith('v_mfma') and (dst & r2):
                    note("%s write -> MFMA read" % op, k, nm, i, (i + k) % n)
                    break
                if dst & w2:
                    break
                nm += op2.startswith('v_mfma')
    for kind, (d_ins, d_mfma, a, b) in sorted(worst.items()):
        print(f"{kind}: closest pair {d_ins} instructions / {d_mfma} MFMAs apart: {a[0]} {', '.join(a[1])}  ->  {b[0]} {', '.join(b[1])}")


if __name__ == "__main__":
    main(*sys.argv[1:5])
