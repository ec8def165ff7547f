"""world_size-2/3 gloo tests (CPU) of the sequence-parallel layer (easyanimate_amd/sequence_parallel.py).

The layer is communication + indexing only, so it runs on CPU tensors.  The per-token arithmetic in these tests is
the ORACLE's (tests may use it); the product plugs its HIP kernels into the same layout.  Checked:
  * token partition / layout arithmetic (ragged N, padded tail only at the end of the sequence),
  * K / V^T all-gather reproduces the full-sequence buffers on every rank,
  * a sharded MMDiT block (local norms/GEMMs, gathered K/V, local queries incl. replicated text rows) equals the
    unsharded oracle block, and gather_tokens returns the full prediction on every rank."""
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


def _worker(rank, world, port, n_video, T, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from easyanimate_amd.sequence_parallel import SequenceParallel
        from easyanimate_amd.synthetic import synth_state_dict
        from oracle import restatement as R
        torch.manual_seed(0)
        torch.set_num_threads(1)
        d, H, B = 128, 2, 2
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

        sp = SequenceParallel()
        hl = sp.shard_tokens(h)
        lo, hi = sp.shard_range()
        assert hl.shape[1] == hi - lo and sp.n_loc % 64 == 0
        cl, sl = sp.shard_rope((cos, sin), "cpu")
        S, q_begin, q_end, off_v, s_pad = sp.layout(T, hl.shape[1])
        assert S == T + n_video and q_begin == T + lo and q_end == T + hi and s_pad % 256 == 0 and s_pad >= T + world * sp.n_loc

        # ---- local per-token work (oracle arithmetic), global K/V layout, exchange
        nh, ne, gate, egate = R.layernorm_zero(sd, "norm1.", hl, e, temb, 1e-6)

        def qkv(pre, x, rope):
            q = F.linear(x, sd[pre + "to_q.weight"], sd[pre + "to_q.bias"]).view(B, -1, H, 64).transpose(1, 2)
            k = F.linear(x, sd[pre + "to_k.weight"], sd[pre + "to_k.bias"]).view(B, -1, H, 64).transpose(1, 2)
            v = F.linear(x, sd[pre + "to_v.weight"], sd[pre + "to_v.bias"]).view(B, -1, H, 64).transpose(1, 2)
            q = F.layer_norm(q, (64,), sd[pre + "norm_q.weight"], sd[pre + "norm_q.bias"], 1e-6)
            k = F.layer_norm(k, (64,), sd[pre + "norm_k.weight"], sd[pre + "norm_k.bias"], 1e-6)
            if rope is not None:
                q, k = R.apply_rotary_emb(q, *rope), R.apply_rotary_emb(k, *rope)
            return q, k, v

        qv, kv, vv = qkv("attn1.", nh, (cl, sl))
        qt, kt, vt_ = qkv("attn2.", ne, None)
        ws = dict(q=torch.zeros(B, H, s_pad, 64), k=torch.zeros(B, H, s_pad, 64), vt=torch.zeros(B, H, 64, s_pad))
        n = hi - lo
        ws["q"][:, :, :T], ws["k"][:, :, :T], ws["vt"][:, :, :, :T] = qt, kt, vt_.transpose(2, 3)
        ws["q"][:, :, off_v:off_v + n], ws["k"][:, :, off_v:off_v + n] = qv, kv
        ws["vt"][:, :, :, off_v:off_v + n] = vv.transpose(2, 3)
        sp.exchange_kv(ws, T, n)

        # every rank now holds the full K / V^T: compare with the unsharded projection
        nh_full, ne_full, _, _ = R.layernorm_zero(sd, "norm1.", h, e, temb, 1e-6)
        _, k_full, v_full = qkv("attn1.", nh_full, (cos, sin))
        assert torch.allclose(ws["k"][:, :, T:T + n_video], k_full, atol=1e-5)
        assert torch.allclose(ws["vt"][:, :, :, T:T + n_video], v_full.transpose(2, 3), atol=1e-5)
        assert ws["k"][:, :, S:].abs().max() == 0  # padding only at the end

        # ---- local queries (text rows + own video rows) over all S keys
        K_, V_ = ws["k"][:, :, :S], ws["vt"][:, :, :, :S].transpose(2, 3)
        o = torch.zeros(B, S, d)
        for (a, b_) in ((0, T), (q_begin, q_end)):
            oo = F.scaled_dot_product_attention(ws["q"][:, :, a:b_], K_, V_)
            o[:, a:b_] = oo.transpose(1, 2).reshape(B, b_ - a, d)
        o_t, o_v = sp.split_output(o, T, n)
        ah = F.linear(o_v, sd["attn1.to_out.0.weight"], sd["attn1.to_out.0.bias"])
        ae = F.linear(o_t, sd["attn2.to_out.0.weight"], sd["attn2.to_out.0.bias"])
        h1 = hl + gate * ah
        e1 = e + egate * ae
        nh, ne, gff, egff = R.layernorm_zero(sd, "norm2.", h1, e1, temb, 1e-6)
        h2 = h1 + gff * R.feed_forward(sd, "ff.", nh)
        e2 = e1 + egff * R.feed_forward(sd, "txt_ff.", ne)

        full = sp.gather_tokens(h2)
        assert full.shape == h_ref.shape
        err_h = (full - h_ref).abs().max().item()
        err_e = (e2 - e_ref).abs().max().item()
        # text stream must be bit-identical across ranks
        e_all = [torch.empty_like(e2) for _ in range(world)]
        dist.all_gather(e_all, e2)
        same = all(torch.equal(e_all[0], x) for x in e_all)
        ret[rank] = (err_h, err_e, same)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_video,T", [(2, 256, 64), (2, 200, 7), (3, 330, 16)])
def test_sequence_parallel_block_equals_unsharded(world, n_video, T):
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), n_video, T, ret), nprocs=world, join=True)
    assert len(ret) == world
    for r in range(world):
        err_h, err_e, same = ret[r]
        assert err_h < 2e-5 and err_e < 2e-5 and same, (r, err_h, err_e, same)


def test_partition_arithmetic():
    from easyanimate_amd.sequence_parallel import SequenceParallel

    class Fake(SequenceParallel):
        def __init__(self, world, rank):
            self.world, self.rank, self.group = world, rank, None
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
