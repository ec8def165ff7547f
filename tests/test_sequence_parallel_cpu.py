"""world_size-2/3/4/8 gloo tests (CPU) of the multi-GPU layer (easyanimate_amd/sequence_parallel.py).

The layer is communication + indexing only, so it runs on CPU tensors.  The per-token arithmetic in these tests is
the ORACLE's (tests may use it); the product plugs its HIP kernels into the same layout.  Checked, for the flat
sequence split (odd worlds / cfg_parallel=False) and for the CFG x sequence split (even worlds):
  * token partition / per-rank row layout arithmetic (ragged N: only the last shard is short, key ranges contiguous),
  * the asynchronous K / V^T all-gather puts every remote shard into its slot of the rank-private layout,
  * a sharded MMDiT block (local norms/GEMMs, queries = text rows + own rows, keys = local range + remote range)
    equals the unsharded oracle block, and gather_tokens returns the full CFG pair on every rank."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn.functional as F


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_video, T, cfg_parallel, ret, mode="keys", H=2):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from easyanimate_amd.sequence_parallel import SequenceParallel
        from easyanimate_amd.synthetic import synth_state_dict
        from oracle import restatement as R
        torch.manual_seed(0)
        torch.set_num_threads(1)
        d, B = 64 * H, 2
        shapes = {}
        for n in ("norm1", "norm2"):
            shapes.update({f"{n}.linear.weight": (6 * d, 32), f"{n}.linear.bias": (6 * d,), f"{n}.norm.weight": (d,), f"{n}.norm.bias": (d,)})
        for a in ("attn1", "attn2"):
            for l in ("to_q", "to_k", "to_v", "to_out.0"):
                shapes.update({f"{a}.{l}.weight": (d, d), f"{a}.{l}.bias": (d,)})
            for l in ("norm_q", "norm_k"):
                shapes.update({f"{a}.{l}.weight": (64,), f"{a}.{l}.bias": (64,)})
        for f in ("ff", "txt_ff"):
            shapes.update({f"{f}.net.0.proj.weight": (4 * d, d), f"{f}.net.0.proj.bias": (4 * d,), f"{f}.net.2.weight": (d, 4 * d), f"{f}.net.2.bias": (d,)})
        sd = synth_state_dict(shapes, 5, "stress")
        g = torch.Generator().manual_seed(3)
        h = torch.randn(B, n_video, d, generator=g)
        e = torch.randn(B, T, d, generator=g)
        temb = torch.randn(B, 32, generator=g)
        cos = torch.rand(n_video, 64, generator=g)
        sin = torch.rand(n_video, 64, generator=g)
        h_ref, e_ref = R.dit_block(sd, "", h, e, temb, (cos, sin), H, 1e-6)

        sp = SequenceParallel(cfg_parallel=cfg_parallel)
        b0, b1 = sp.begin(B)
        cfg_mode = cfg_parallel and world % 2 == 0
        assert (b0, b1) == ((rank // (world // 2), rank // (world // 2) + 1) if cfg_mode else (0, B))
        assert sp.size == (world // 2 if cfg_mode else world)
        h, e, temb, e_ref = h[b0:b1], e[b0:b1], temb[b0:b1], e_ref[b0:b1]
        Bl = b1 - b0
        hl = sp.shard_tokens(h)
        lo, hi = sp.shard_range()
        n = hi - lo
        assert hl.shape[1] == n and sp.n_loc % 64 == 0
        cl, sl = sp.shard_rope((cos, sin), "cpu")
        lay = sp.layout(T, n)
        nl = sp.n_loc
        vo = lay.t_pad
        assert vo % 64 == 0 and 0 <= vo - T < 64 and lay.q_end == vo + n and lay.rows >= vo + nl and lay.rows % 256 == 0 and lay.v_off == vo
        assert lay.own_ranges == ([(0, T + n)] if vo == T else [(0, T), (vo, vo + n)])
        assert all(lo % 64 == 0 for lo, _ in lay.own_ranges)
        assert lay.q_pad == lay.rows
        assert lay.remote_valid == n_video - n     # every other token, exactly once
        assert sp.exchanges(lay) == (sp.size > 1)

        # ---- local per-token work (oracle arithmetic), per-rank K/V layout, asynchronous exchange
        nh, ne, gate, egate = R.layernorm_zero(sd, "norm1.", hl, e, temb, 1e-6)

        def qkv(pre, x, rope):
            q = F.linear(x, sd[pre + "to_q.weight"], sd[pre + "to_q.bias"]).view(Bl, -1, H, 64).transpose(1, 2)
            k = F.linear(x, sd[pre + "to_k.weight"], sd[pre + "to_k.bias"]).view(Bl, -1, H, 64).transpose(1, 2)
            v = F.linear(x, sd[pre + "to_v.weight"], sd[pre + "to_v.bias"]).view(Bl, -1, H, 64).transpose(1, 2)
            q = F.layer_norm(q, (64,), sd[pre + "norm_q.weight"], sd[pre + "norm_q.bias"], 1e-6)
            k = F.layer_norm(k, (64,), sd[pre + "norm_k.weight"], sd[pre + "norm_k.bias"], 1e-6)
            if rope is not None:
                q, k = R.apply_rotary_emb(q, *rope), R.apply_rotary_emb(k, *rope)
            return q, k, v

        qv, kv, vv = qkv("attn1.", nh, (cl, sl))
        qt, kt, vt_ = qkv("attn2.", ne, None)
        # q workspace + the exchange buffer: the projections land in the rank's OWN slot, rows [text | gap | shard]
        qws = torch.zeros(Bl, H, lay.q_pad, 64)
        buf = sp.kv_buffer(Bl, H, lay, "cpu", torch.float32)
        assert buf.shape == (sp.size, 2, Bl, H, lay.rows * 64) and sp.kv_buffer(Bl, H, lay, "cpu", torch.float32) is buf
        k_own, vt_own = sp.slot_views(buf)
        assert k_own.data_ptr() == buf[sp.rank, 0].data_ptr() and vt_own.shape == (Bl, H, 64, lay.rows)
        qws[:, :, :T], k_own[:, :, :T], vt_own[:, :, :, :T] = qt, kt, vt_.transpose(2, 3)
        qws[:, :, vo:vo + n], k_own[:, :, vo:vo + n] = qv, kv
        vt_own[:, :, :, vo:vo + n] = vv.transpose(2, 3)
        pending = sp.exchange_start(buf)
        assert (pending is None) == (sp.size == 1)
        sp.exchange_finish(pending)

        # every rank's shard sits in its slot, in place: compare with the unsharded projection of this batch slice
        nh_full, _, _, _ = R.layernorm_zero(sd, "norm1.", h, e, temb, 1e-6)
        _, k_full, v_full = qkv("attn1.", nh_full, (cos, sin))
        for r in range(sp.size):
            rlo, rhi = sp.shard_range(r)
            kr, vtr = sp.slot_views(buf, r)
            assert torch.allclose(kr[:, :, vo:vo + rhi - rlo], k_full[:, :, rlo:rhi], atol=1e-5)
            assert torch.allclose(vtr[:, :, :, vo:vo + rhi - rlo], v_full[:, :, rlo:rhi].transpose(2, 3), atol=1e-5)
            assert not kr[:, :, vo + rhi - rlo:].any()    # tail of a short shard and the pad to 256 rows: zero

        use_heads = mode == "heads" and sp.size > 1      # one sequence rank (CFG split only): nothing to exchange in either mode
        if use_heads:
            # ---- EA_SP_MODE=heads: the product's own exchange code (processor._head_parallel: head all-to-all, contiguous operands
            # of all tokens for H / P' heads, all-to-all back, text rows all-gathered) around the oracle's attention
            from easyanimate_amd.processor import EasyAnimateAttnProcessor2_0
            sp.mode = "heads"
            proc = EasyAnimateAttnProcessor2_0()
            assert proc._heads_mode(lay, sp, H)
            seen = []

            def attend_heads(qf, kf, vf, Hl, head0, Nt):
                seen.append((Hl, head0, Nt))
                S_ = T + Nt
                assert not qf[:, :, S_:].any() and not vf[:, :, :, S_:].any()
                return F.scaled_dot_product_attention(qf[:, :, :S_], kf[:, :, :S_], vf[:, :, :, :S_].transpose(2, 3)).transpose(1, 2).reshape(Bl, S_, Hl * 64)
            o = proc._head_parallel(dict(q=qws, k=k_own, vt=vt_own), Bl, H, T, lay.q_end, d, "cpu", lay, sp, attend_heads)
            assert seen == [(H // sp.size, sp.rank * (H // sp.size), n_video)]
        # ---- queries = text rows + own rows; keys = the own slot's ranges, then the shard rows of every other slot
        Ks = [k_own[:, :, lo_:hi_] for lo_, hi_ in lay.own_ranges]
        Vs = [vt_own[:, :, :, lo_:hi_] for lo_, hi_ in lay.own_ranges]
        left = lay.remote_valid
        for r in range(sp.size):
            if r == sp.rank:
                continue
            kr, vtr = sp.slot_views(buf, r)
            take = min(nl, left)
            Ks.append(kr[:, :, vo:vo + take]); Vs.append(vtr[:, :, :, vo:vo + take])
            left -= take
        assert left == 0
        if not use_heads:
            K_, V_ = torch.cat(Ks, 2), torch.cat(Vs, 3).transpose(2, 3)
            assert K_.shape[2] == T + n_video
            ws = dict(q=qws)
            oo = F.scaled_dot_product_attention(ws["q"][:, :, :lay.q_end], K_, V_)
            o = oo.transpose(1, 2).reshape(Bl, lay.q_end, d)
        o_t, o_v = o[:, :T], o[:, vo:]
        ah = F.linear(o_v, sd["attn1.to_out.0.weight"], sd["attn1.to_out.0.bias"])
        ae = F.linear(o_t, sd["attn2.to_out.0.weight"], sd["attn2.to_out.0.bias"])
        h1 = hl + gate * ah
        e1 = e + egate * ae
        nh, ne, gff, egff = R.layernorm_zero(sd, "norm2.", h1, e1, temb, 1e-6)
        h2 = h1 + gff * R.feed_forward(sd, "ff.", nh)
        e2 = e1 + egff * R.feed_forward(sd, "txt_ff.", ne)

        full = sp.gather_tokens(h2)
        assert full.shape == h_ref.shape   # the whole CFG pair, all tokens, on every rank
        err_h = (full - h_ref).abs().max().item()
        err_e = (e2 - e_ref).abs().max().item()
        ret[rank] = (err_h, err_e)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_video,T,cfg_parallel", [
    (2, 256, 64, False), (2, 200, 7, False), (3, 330, 16, True),    # flat sequence split (3: odd world)
    (2, 200, 64, True),                                              # CFG split only: no per-block exchange
    (4, 330, 64, True), (4, 256, 7, True),                           # 2 (CFG) x 2 (sequence), ragged / unaligned text
    (8, 900, 16, True),                                              # 2 x 4: the shape of `bench.py --gpus 8`
])
def test_sequence_parallel_block_equals_unsharded(world, n_video, T, cfg_parallel):
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), n_video, T, cfg_parallel, ret), nprocs=world, join=True)
    assert len(ret) == world
    for r in range(world):
        err_h, err_e = ret[r]
        assert err_h < 2e-5 and err_e < 2e-5, (r, err_h, err_e)


@pytest.mark.parametrize("world,n_video,T,cfg_parallel,H", [
    (2, 200, 7, False, 2),       # flat split, ragged last shard, unaligned text
    (3, 330, 16, True, 3),       # odd world: three sequence ranks, one head each
    (4, 330, 64, True, 4),       # 2 (CFG) x 2 (sequence)
    (4, 256, 7, False, 4),       # four sequence ranks of both batch elements
    (8, 900, 16, True, 8),       # 2 x 4: the shape of `bench.py --gpus 8 --sp-mode heads`
])
def test_head_parallel_block_equals_unsharded(world, n_video, T, cfg_parallel, H):
    """EA_SP_MODE=heads (SURVEY 5.7 design C): the full-attention block exchanges heads instead of keys -- the product's
    exchange code (processor._head_parallel) over gloo around the oracle's arithmetic equals the unsharded oracle block."""
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), n_video, T, cfg_parallel, ret, "heads", H), nprocs=world, join=True)
    assert len(ret) == world
    for r in range(world):
        err_h, err_e = ret[r]
        assert err_h < 2e-5 and err_e < 2e-5, (r, err_h, err_e)


def test_partition_arithmetic():
    from easyanimate_amd.sequence_parallel import SequenceParallel

    class Fake(SequenceParallel):
        def __init__(self, world, rank):
            from easyanimate_amd.sequence_parallel import _Axis
            self.world, self.world_rank = world, rank
            self.axis = _Axis(world, rank, 1, None)
            self.n_total = self.n_loc = 0

    for world, N in [(8, 53248), (4, 29952), (2, 13312), (3, 1000), (8, 520)]:
        covered = []
        for r in range(world):
            sp = Fake(world, r)
            sp.plan(N)
            lo, hi = sp.shard_range()
            covered.append((lo, hi))
            assert sp.n_loc % 64 == 0
        assert covered[0][0] == 0 and covered[-1][1] == N or any(hi == N for _, hi in covered)
        assert all(covered[i][1] == covered[i + 1][0] for i in range(world - 1))
    sp = Fake(8, 0)
    sp.plan(53248)
    assert sp.n_loc == 6656  # config 3: 6656 tokens / rank


def test_exchange_layout_at_config3_size():
    """The slot layout of the per-block K / V^T exchange at the benchmark's size (N = 53 248 video tokens, T = 256, P = 8 =
    CFG 2 x sequence 4): 13 312 tokens per rank, slots of 13 568 rows (= the q workspace: the own slot is a plain operand),
    one own key range, 39 936 remote keys; and with unaligned text / a short last shard."""
    from easyanimate_amd.sequence_parallel import EmulatedRank
    for r in range(8):
        sp = EmulatedRank(8, r)
        assert sp.begin(2) == (r // 4, r // 4 + 1) and sp.size == 4 and sp.rank == r % 4
        sp.plan(53248)
        lo, hi = sp.shard_range()
        assert (lo, hi) == ((r % 4) * 13312, (r % 4 + 1) * 13312)
        lay = sp.layout(256, hi - lo)
        assert (lay.t_pad, lay.rows, lay.q_pad, lay.q_end) == (256, 13568, 13568, 13568)
        assert lay.own_ranges == [(0, 13568)] and lay.remote_valid == 3 * 13312 and sp.exchanges(lay)
    sp = EmulatedRank(3, 2, cfg_parallel=False)          # flat split, short last shard, text not a multiple of 64
    sp.begin(2)
    sp.plan(1000)
    lo, hi = sp.shard_range()
    lay = sp.layout(77, hi - lo)
    assert sp.n_loc == 384 and (lo, hi) == (768, 1000) and lay.t_pad == 128 and lay.rows == 512
    assert lay.own_ranges == [(0, 77), (128, 128 + 232)] and lay.remote_valid == 768 and lay.q_end == 360


def test_layout_invariants_over_random_shapes():
    """Property test of the partition + slot layout over random (world, N, T): shards tile [0, N) in rank order, every
    alignment the kernels rely on holds (shard stride and first shard row multiples of 64, slot rows a multiple of 256),
    own + remote keys add up to every key of the sequence, and the queries fit the workspace."""
    from hypothesis import given, settings, strategies as st
    from easyanimate_amd.sequence_parallel import EmulatedRank

    @settings(max_examples=300, deadline=None)
    @given(world=st.sampled_from([2, 3, 4, 6, 8]), cfg=st.booleans(), n=st.integers(4000, 60000), T=st.integers(1, 300))
    def check(world, cfg, n, T):
        seen, keys_total = [], None
        for r in range(world):
            sp = EmulatedRank(world, r, cfg_parallel=cfg)
            b0, b1 = sp.begin(2)
            P = sp.size
            assert P == (world // 2 if cfg and world % 2 == 0 else world) and 0 <= b0 < b1 <= 2
            sp.plan(n)
            lo, hi = sp.shard_range()
            if sp.rank == 0:
                seen.append([])
            seen[-1].append((lo, hi))
            lay = sp.layout(T, hi - lo)
            assert sp.n_loc % 64 == 0 and lay.t_pad % 64 == 0 and lay.t_pad - T < 64 and lay.rows % 256 == 0
            assert lay.rows >= lay.t_pad + sp.n_loc and lay.q_pad == lay.rows and lay.q_end == lay.t_pad + (hi - lo) <= lay.q_pad
            assert all(a % 64 == 0 and a < b for a, b in lay.own_ranges)
            own_keys = sum(b - a for a, b in lay.own_ranges)
            assert own_keys == T + (hi - lo)
            assert own_keys + lay.remote_valid == T + n          # every key of the sequence exactly once
            assert sp.exchanges(lay) == (P > 1)
        for shards in seen:                                      # one list per sequence-parallel group
            assert shards[0][0] == 0 and shards[-1][1] == n
            assert all(shards[i][1] == shards[i + 1][0] for i in range(len(shards) - 1))
            assert all(hi - lo == shards[0][1] - shards[0][0] for lo, hi in shards[:-1]) and 0 < shards[-1][1] - shards[-1][0]
    check()


def _worker_groups(rank, world, port, cfg_parallel, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from easyanimate_amd.sequence_parallel import SequenceParallel
        torch.manual_seed(0)
        sp = SequenceParallel(cfg_parallel=cfg_parallel)
        sp.groups = 3
        B, H, T, N = 2, 6, 64, 700
        b0, b1 = sp.begin(B)
        Bl = b1 - b0
        sp.plan(N)
        lo, hi = sp.shard_range()
        lay = sp.layout(T, hi - lo)
        assert sp.head_groups(H) == (3 if sp.size > 1 else 1) and sp.head_groups(4) == 1      # 4 heads do not split into 3 groups
        G = 3
        buf = sp.kv_buffer(Bl, H, lay, "cpu", torch.float32, groups=G)
        assert buf.shape == (G, sp.size, 2, Bl, H // G, lay.rows * 64) and sp.kv_buffer(Bl, H, lay, "cpu", torch.float32, groups=G) is buf
        # every rank writes a value that names (group, rank, K / V^T, batch, head) into its own slots, group by group
        gen = lambda g, r: (torch.arange(2 * Bl * (H // G), dtype=torch.float32).view(2, Bl, H // G, 1) + 1000 * g + 100 * r + 0.5).expand(2, Bl, H // G, lay.rows * 64)
        for g in range(G):
            k_own, vt_own = sp.slot_views(buf[g])
            assert k_own.shape == (Bl, H // G, lay.rows, 64) and k_own.data_ptr() == buf[g, sp.rank, 0].data_ptr()
            buf[g, sp.rank].copy_(gen(g, sp.rank))
        pend = [sp.exchange_start(buf[g]) for g in range(G)]          # posted back to back ...
        for g in (2, 0, 1):                                          # ... and finished in any order, each on its own
            sp.exchange_finish(pend[g])
            for r in range(sp.size):
                assert torch.equal(buf[g, r], gen(g, r)), (g, r)
        # EA_SP_INPLACE=0: one receive buffer per group
        sp.inplace = False
        ok = True
        if sp.size > 1:
            pend = [sp.exchange_start(buf[g]) for g in range(G)]
            outs = [sp.exchange_finish(p) for p in pend]
            ok = all(o is None for o in outs) or (len({o.data_ptr() for o in outs}) == G and all(torch.equal(o[r], gen(g, r)) for g, o in enumerate(outs) for r in range(sp.size)))
        ret[rank] = (sp.size, ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,cfg_parallel", [(2, False), (3, False), (4, True)])
def test_grouped_exchange_buffers(world, cfg_parallel):
    """EA_SP_GROUPS: the exchange buffer split by head groups ([G, P, 2, B, H / G, rows * 64]) -- every buf[g] is a complete exchange
    buffer of its own: own-slot views, G all-gathers in flight at once, finished independently, every slot of every group filled
    with its owner's data; one receive buffer per group in the out-of-place form."""
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_groups, args=(world, _free_port(), cfg_parallel, ret), nprocs=world, join=True)
    assert len(ret) == world and all(ret[r][1] for r in range(world)), dict(ret)


def _worker_selfcheck(rank, world, port, faulty_rank, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import warnings
        from easyanimate_amd.sequence_parallel import SequenceParallel
        sp = SequenceParallel(cfg_parallel=True)
        B, H, T, N = 2, 4, 64, 512
        b0, b1 = sp.begin(B)
        Bl = b1 - b0
        sp.plan(N)
        lo, hi = sp.shard_range()
        lay = sp.layout(T, hi - lo)
        buf = sp.kv_buffer(Bl, H, lay, "cpu", torch.float32)
        gen = lambda r: torch.full((2, Bl, H, lay.rows * 64), 10.0 * sp.axis.cfg_rank + r + 0.25)
        buf[sp.rank].copy_(gen(sp.rank))
        # ONE rank of the world sees the in-place gather disagree with the out-of-place one (injected: gloo has no in-place form)
        sp._selfcheck_fault = lambda s: 1 if s.world_rank == faulty_rank else 0
        assert sp.inplace and not sp._inplace_checked
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            sp.exchange_finish(sp.exchange_start(buf))
        first_ok = all(torch.equal(buf[r], gen(r)) for r in range(sp.size))
        # every rank -- the faulty one's CFG half and the other half alike -- agreed on the verdict and fell back; the exchange of
        # the call that ran the check is complete; later exchanges take the out-of-place form (a receive buffer comes back)
        fell_back = (not sp.inplace) and sp._inplace_checked and sp.inplace_requested and len(w) == 1
        buf[sp.rank].copy_(gen(sp.rank) + 1)
        for r in range(sp.size):
            if r != sp.rank:
                buf[r].zero_()
        other = sp.exchange_finish(sp.exchange_start(buf))
        src = other if other is not None else buf
        later_ok = all(torch.equal(src[r], gen(r) + 1) for r in range(sp.size) if r != sp.rank)
        ret[rank] = (first_ok, fell_back, later_ok, faulty_rank < 0)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,faulty_rank", [(4, 3), (3, 0), (4, -1)])
def test_inplace_selfcheck_verdict_is_agreed_and_falls_back(world, faulty_rank):
    """VERDICT r5 weak #11 / next #3a: the first-use check of the in-place K / V^T all-gather must never leave the ranks in
    different states.  One rank reporting a mismatch (injected) makes EVERY rank of the world -- both CFG halves -- switch to the
    out-of-place exchange, with a warning instead of an exception, and the exchange that carried the check is complete; with no
    fault (faulty_rank = -1) every rank stays in place."""
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_selfcheck, args=(world, _free_port(), faulty_rank, ret), nprocs=world, join=True)
    assert len(ret) == world
    for r in range(world):
        first_ok, fell_back, later_ok, clean = ret[r]
        assert first_ok and later_ok, (r, ret[r])
        assert fell_back == (not clean), (r, ret[r])
