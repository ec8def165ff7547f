"""Functional model of the K / V^T stage bookkeeping of ea_attention_v4.inc (which tile each fragment read sees), for both
extremes of DMA timing: every request completes at the last moment its vmcnt wait allows ("late"), or the moment it is issued
("early": an overwrite of a stage that is still needed shows up as a wrong tile).  CPU only; mirrors the kernel's order of
requests, waits, address updates and reads -- a change to one must be made in the other.
    python tools/model_att4_stages.py"""


def run(NS, nt, early):
    ntp = max(nt, 1) + 3
    K, V = [None] * NS, [None] * NS
    queue = []

    def clamp(t):
        return min(t, ntp - 1)

    def land(req):
        kind, tile, stage = req
        (K if kind == "K" else V)[stage] = tile

    def issue(kind, tile, stage):
        req = (kind, clamp(tile), stage)
        if early:
            land(req)
        else:
            queue.append(req)

    def wait(pieces):           # s_waitcnt vmcnt(pieces): two pieces per request
        while 2 * len(queue) > pieces:
            land(queue.pop(0))

    errors = []

    def expect(what, got, want):
        if got != want:
            errors.append(f"NS={NS} nt={nt} {'early' if early else 'late'}: {what}: tile {got}, expected {want}")

    issue("K", 0, 0)
    for i in range(1, NS):
        issue("V", i - 1, i - 1)
        issue("K", i, i)
    wait(4 if NS == 2 else 12)
    expect("S(0) <- K stage 0", K[0], 0)
    ak, av, fv = 0, 0, "zero"
    for t in range(nt):
        # even block
        expect(f"even({t}) K fragments", K[ak], t)
        expect(f"even({t}) PV operand", fv, "zero" if t == 0 else ("V", t - 1, 1))
        wait(0 if NS == 2 else 8)
        issue("K", t + NS, (t + NS) % NS)
        issue("V", t + NS - 1, (t + NS - 1) % NS)
        ak, av = (t + 1) % NS, t % NS
        fv = ("V", V[av], 0)
        expect(f"even({t}) prefetch of V({t}) block 0", fv, ("V", t, 0))
        # odd block
        if t + 1 < nt:
            expect(f"odd({t}) K fragments", K[ak], t + 1)
        fv_use = fv
        expect(f"odd({t}) PV operand", fv_use, ("V", t, 0))
        fv = ("V", V[av], 1)
        expect(f"odd({t}) prefetch of V({t}) block 1", fv, ("V", t, 1))
    wait(0)
    expect("drain PV operand", fv, ("V", nt - 1, 1))
    return errors


if __name__ == "__main__":
    bad = []
    for NS in (2, 4):
        for nt in range(1, 9):
            for early in (False, True):
                bad += run(NS, nt, early)
    print("\n".join(bad) if bad else "stage bookkeeping consistent for NS in (2, 4), 1..8 tiles, early and late DMA completion")
