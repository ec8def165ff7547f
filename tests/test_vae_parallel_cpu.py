"""Temporal-parallel VAE (easyanimate_amd/vae_parallel.py) on CPU with gloo: the partition, the per-convolution halo
exchange and the dropped-output rule, applied to the ORACLE's causal convolutions (F.conv3d arithmetic), must reproduce the
whole-clip evaluation exactly -- stride 1, stride 2 (encoder down-samplers), the temporal duplication of the decoder's
up-samplers, ranks with different frame counts, and more ranks than can be active."""
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


def _conv(x, w, stride=1):
    """whole-clip causal convolution, frames first: x [T, C, H, W] (oracle/restatement_vae.causal_conv3d arithmetic)"""
    xx = x.permute(1, 0, 2, 3)[None]
    xx = F.pad(xx, (0, 0, 0, 0, 2, 0), mode="replicate")
    return F.conv3d(xx, w, stride=(stride, 1, 1), padding=(0, 1, 1))[0].permute(1, 0, 2, 3)


def _tdup(y):
    return y if y.shape[0] == 1 else torch.cat([y[:1], y[1:].repeat_interleave(2, dim=0)])


def _split_conv(tp, x, w, stride=1, tdup=False):
    """what vae_modules.conv_cl does under a temporal split, with the oracle convolution in place of the kernel"""
    n = tp.halo_frames(stride)
    halo = tp.exchange(x, n)
    if halo is None:
        y = _conv(x, w, stride)
        return _tdup(y) if tdup else y
    y = _conv(torch.cat([halo, x]), w, stride)
    if tdup:
        y = _tdup(y)
    return y[tp.dropped_outputs(stride, n, tdup):]


def _weights():
    g = torch.Generator().manual_seed(3)
    return [torch.randn(4, 4, 3, 3, 3, generator=g, dtype=torch.float64) / 6 for _ in range(8)]


def _encode(conv, x, w):    # 49-frame level -> stride 2 -> stride 2 (as Encoder: two temporal down-samplers)
    x = torch.tanh(conv(x, w[0]))
    x = torch.tanh(conv(x, w[1], 2))
    x = torch.tanh(conv(x, w[2]))
    x = torch.tanh(conv(x, w[3], 2))
    return conv(x, w[4])


def _decode(conv, z, w):    # latent level -> x2 -> x2 in time (as Decoder: two temporal up-samplers)
    z = torch.tanh(conv(z, w[5]))
    z = torch.tanh(conv(z, w[6], 1, True))
    z = torch.tanh(conv(z, w[7], 1, True))
    return conv(z, w[0])


def _worker(rank, world, port, frames, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from easyanimate_amd.vae_parallel import TemporalParallel
        w = _weights()
        g = torch.Generator().manual_seed(7)
        video = torch.randn(frames, 4, 6, 5, generator=g, dtype=torch.float64)
        t_lat = (frames - 1) // 4 + 1
        z = torch.randn(t_lat, 4, 6, 5, generator=g, dtype=torch.float64)
        full = lambda x, ww, s=1, d=False: (_tdup(_conv(x, ww, s)) if d else _conv(x, ww, s))
        ref_e, ref_d = _encode(full, video, w), _decode(full, z, w)
        assert ref_e.shape[0] == t_lat and ref_d.shape[0] == frames
        tp = TemporalParallel()
        ranges = tp.plan(t_lat)
        assert ranges[0][0] == 0 and ranges[-1][1] == t_lat and all(b - a >= 2 or b == a for a, b in ranges)
        assert tp.active_ranks == min(world, t_lat // 2)
        fr = tp.finer(tp.finer(ranges[rank]))
        split = lambda x, ww, s=1, d=False: _split_conv(tp, x, ww, s, d)
        e_loc = _encode(split, video[fr[0]:fr[1]], w) if tp.is_active else None
        d_loc = _decode(split, z[ranges[rank][0]:ranges[rank][1]], w) if tp.is_active else None
        if tp.is_active:
            assert e_loc.shape[0] == ranges[rank][1] - ranges[rank][0] and d_loc.shape[0] == fr[1] - fr[0]
        e = tp.gather_frames(e_loc, ranges, 0, ref_e[:1])
        d = tp.gather_frames(d_loc, [tp.finer(tp.finer(r)) for r in ranges], 0, ref_d[:1])
        ret[rank] = ((e - ref_e).abs().max().item(), (d - ref_d).abs().max().item(), tp.messages, tp.active_ranks)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,frames", [(2, 49), (3, 49), (4, 17), (3, 9), (2, 5)])
def test_temporal_split_equals_whole_clip(world, frames):
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), frames, ret), nprocs=world, join=True)
    assert len(ret) == world
    for r in range(world):
        err_e, err_d, msgs, active = ret[r]
        assert err_e < 1e-12 and err_d < 1e-12, (r, ret[r])
    # every active rank but the first received one halo per causal convolution (5 in the encoder, 4 in the decoder)
    active = ret[0][3]
    assert ret[0][2] == 0 and all(ret[r][2] == 9 for r in range(1, active)) and all(ret[r][2] == 0 for r in range(active, world))


def test_partition_and_level_mapping():
    from easyanimate_amd.vae_parallel import TemporalParallel
    f = TemporalParallel.finer
    assert f((0, 2)) == (0, 3) and f((2, 4)) == (3, 7) and f(f((2, 4))) == (5, 13) and f(f((0, 13))) == (0, 49)
    assert TemporalParallel.dropped_outputs(1, 2, False) == 2 and TemporalParallel.dropped_outputs(1, 2, True) == 3
    assert TemporalParallel.dropped_outputs(2, 1, False) == 1


# ---- space x time: the row split composed with the temporal split (8 GPUs on 13 latent frames = 4 x 2) -------------------
def _conv_hw(x, w, stride=1, sstride=1, ups=False):
    """whole-clip causal convolution with the VAE's spatial conventions: pad 1 (stride 1), or pad 0 + one zero row / column on
    the high side (stride 2, downsamplers.py:44-46); ups = nearest x2 in space first.  x [T, C, H, W]."""
    xx = x.permute(1, 0, 2, 3)[None]
    if ups:
        xx = F.interpolate(xx, scale_factor=(1, 2, 2), mode="nearest")
    if sstride == 2:
        xx = F.pad(xx, (0, 1, 0, 1))
    xx = F.pad(xx, (0, 0, 0, 0, 2, 0), mode="replicate")
    return F.conv3d(xx, w, stride=(stride, sstride, sstride), padding=(0, 0 if sstride == 2 else 1, 0 if sstride == 2 else 1))[0].permute(1, 0, 2, 3)


def _gn(x, groups=2, eps=1e-6):
    """per-frame GroupNorm, frames first: x [T, C, H, W]"""
    return F.group_norm(x, groups, eps=eps)


def _grid_conv(tp, x, w, stride=1, sstride=1, ups=False, tdup=False):
    """what vae_modules.conv_cl does under the space x time split (rows = dim 2 here), oracle convolution inside"""
    n = tp.halo_frames(stride)
    halo = tp.exchange(x, n)
    drop = 0
    if halo is not None:
        x = torch.cat([halo, x])
        drop = tp.dropped_outputs(stride, n, tdup)
    keep = None
    if tp.ps > 1:
        h_loc = x.shape[2]
        xr = x.permute(0, 2, 3, 1)        # exchange_rows works on [T, H, W, C]
        above, below = tp.exchange_rows(xr, 0 if sstride == 2 else 1, 1)
        ra = 0 if above is None else above.shape[1]
        x = torch.cat([t for t in (above, xr, below) if t is not None], dim=1).permute(0, 3, 1, 2)
        # the oracle pads both outer edges itself; at an INNER edge the halo row takes the pad's place, so the extra output
        # rows the halo produces are dropped
        if sstride == 1:
            f = 2 if ups else 1
            keep = (f * ra, f * h_loc)
    y = _conv_hw(x, w, stride, sstride, ups)
    if tdup:
        y = _tdup(y)
    y = y[drop:]
    if keep is not None:
        y = y[:, :, keep[0]:keep[0] + keep[1]]
    elif tp.ps > 1 and sstride == 2:
        y = y[:, :, :h_loc // 2]
    return y


def _grid_gn(tp, x, groups=2, eps=1e-6):
    """the statistics half of GroupNorm all-reduced over the ranks of a frame (vae_modules._gn under a spatial split)"""
    T, C = x.shape[0], x.shape[1]
    xg = x.reshape(T, groups, -1).double()
    sums = torch.stack([xg.sum(-1), (xg * xg).sum(-1)], -1)
    sums = tp.all_reduce_rows(sums)
    n = xg.shape[-1] * tp.ps
    mean = sums[..., 0] / n
    rstd = 1.0 / torch.sqrt((sums[..., 1] / n - mean * mean).clamp_min(0) + eps)
    return ((xg - mean[..., None]) * rstd[..., None]).reshape(x.shape).to(x.dtype)


def _grid_attn(tp, x):
    """per-frame single-head attention over all H x W tokens: queries = own rows, keys = the all-gathered frame"""
    T, C, H, W = x.shape
    tok = x.permute(0, 2, 3, 1).reshape(T, H * W, C)
    keys = tp.all_gather_rows(tok, 1) if tp is not None else tok
    o = F.scaled_dot_product_attention(tok, keys, keys)
    return x + o.reshape(T, H, W, C).permute(0, 3, 1, 2)


def _net(conv, gn, attn, x, w, decode):
    if decode:      # mid attention, then x2 space, x2 space + time (twice), like Decoder
        x = conv(x, w[0])
        x = attn(gn(x))
        x = torch.tanh(gn(conv(x, w[1], ups=True)))
        x = torch.tanh(gn(conv(x, w[2], ups=True, tdup=True)))
        x = torch.tanh(gn(conv(x, w[3], ups=True, tdup=True)))
        return conv(x, w[4])
    x = torch.tanh(gn(conv(x, w[5])))                       # Encoder: spatial down, two space-time downs, mid attention
    x = torch.tanh(gn(conv(x, w[6], sstride=2)))
    x = torch.tanh(gn(conv(x, w[7], stride=2, sstride=2)))
    x = torch.tanh(gn(conv(x, w[0], stride=2, sstride=2)))
    x = attn(gn(x))
    return conv(x, w[1])


def _worker_grid(rank, world, port, frames, spatial, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from easyanimate_amd.vae_parallel import TemporalParallel
        w = _weights()
        g = torch.Generator().manual_seed(11)
        Hp, Wp = 32, 24
        video = torch.randn(frames, 4, Hp, Wp, generator=g, dtype=torch.float64)
        t_lat = (frames - 1) // 4 + 1
        z = torch.randn(t_lat, 4, Hp // 8, Wp // 8, generator=g, dtype=torch.float64)
        full_conv = lambda x, ww, stride=1, sstride=1, ups=False, tdup=False: (_tdup(_conv_hw(x, ww, stride, sstride, ups)) if tdup else _conv_hw(x, ww, stride, sstride, ups))
        ref_e = _net(full_conv, _gn, lambda x: _grid_attn(None, x), video, w, False)
        ref_d = _net(full_conv, _gn, lambda x: _grid_attn(None, x), z, w, True)
        assert ref_e.shape == (t_lat, 4, Hp // 8, Wp // 8) and ref_d.shape == (frames, 4, Hp, Wp)
        tp = TemporalParallel(spatial=spatial)
        ranges = tp.plan(t_lat)
        assert len(ranges) == world and ranges[rank] == ranges[(rank // spatial) * spatial]
        assert tp.active_ranks == min(world // spatial, t_lat // 2) and (tp.rank_t, tp.rank_s) == divmod(rank, spatial)
        fr = tp.finer(tp.finer(ranges[rank]))
        conv = lambda x, ww, stride=1, sstride=1, ups=False, tdup=False: _grid_conv(tp, x, ww, stride, sstride, ups, tdup)
        gn = lambda x: _grid_gn(tp, x)
        attn = lambda x: _grid_attn(tp, x)
        e_loc = d_loc = None
        if tp.is_active:
            r0, r1 = tp.rows(Hp)
            e_loc = _net(conv, gn, attn, video[fr[0]:fr[1], :, r0:r1], w, False)
            l0, l1 = tp.rows(Hp // 8)
            d_loc = _net(conv, gn, attn, z[ranges[rank][0]:ranges[rank][1], :, l0:l1], w, True)
            assert e_loc.shape == (ranges[rank][1] - ranges[rank][0], 4, (Hp // 8) // spatial, Wp // 8)
            assert d_loc.shape == (fr[1] - fr[0], 4, Hp // spatial, Wp)
        e = tp.gather_frames(e_loc, ranges, 0, ref_e[:1], row_dim=2)
        d = tp.gather_frames(d_loc, [tp.finer(tp.finer(r)) for r in ranges], 0, ref_d[:1], row_dim=2)
        ret[rank] = ((e - ref_e).abs().max().item(), (d - ref_d).abs().max().item(), tp.messages, tp.row_messages, tp.active_ranks)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,frames,spatial", [(2, 9, 2), (4, 17, 2), (8, 49, 2), (4, 9, 4), (6, 25, 2)])
def test_space_time_split_equals_whole_clip(world, frames, spatial):
    """8 ranks on 49 frames (13 latent) = 4 temporal ranges x 2 row halves: every rank is active (the pure temporal split
    keeps 6 of 8 busy).  Exact on the oracle arithmetic up to the fp64 order of the GroupNorm sums."""
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_grid, args=(world, _free_port(), frames, spatial, ret), nprocs=world, join=True)
    assert len(ret) == world
    active = ret[0][4]
    if (world, frames, spatial) == (8, 49, 2):
        assert active == 4      # all eight ranks busy
    for r in range(world):
        err_e, err_d, msgs, row_msgs, _ = ret[r]
        assert err_e < 1e-10 and err_d < 1e-10, (r, ret[r])
        if r // spatial < active:
            assert row_msgs > 0


def _worker_self_loop(rank, world, port, frames, virtual, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from easyanimate_amd.vae_parallel import SelfLoopTemporal
        w = _weights()
        g = torch.Generator().manual_seed(7)
        t_lat = (frames - 1) // 4 + 1
        z = torch.randn(t_lat, 4, 6, 5, generator=g, dtype=torch.float64)
        full = lambda x, ww, s=1, d=False: (_tdup(_conv(x, ww, s)) if d else _conv(x, ww, s))
        ref_d = _decode(full, z, w)
        tp = SelfLoopTemporal(virtual=virtual)
        ranges = tp.plan(t_lat)
        split = lambda x, ww, s=1, d=False: _split_conv(tp, x, ww, s, d)
        parts = []
        for v in range(tp.active_ranks):
            tp.enter(v)
            parts.append(_decode(split, z[ranges[v][0]:ranges[v][1]], w))
        d = torch.cat(parts)
        ret[rank] = ((d - ref_d).abs().max().item(), tp.messages, tp.active_ranks)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("frames,virtual", [(25, 3), (49, 2), (17, 2)])
def test_self_loop_bring_up_mode_equals_whole_clip(frames, virtual):
    """SelfLoopTemporal: one rank plays the temporal ranks in turn; the tails a virtual rank leaves are consumed by the next in
    convolution order.  (gloo cannot send to itself: here the halo is handed over directly; over RCCL it is a real
    ncclSend / ncclRecv pair, tests/test_vae_parallel_gpu.py::test_rccl_point_to_point_world_of_one.)"""
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_self_loop, args=(1, _free_port(), frames, virtual, ret), nprocs=1, join=True)
    err, msgs, active = ret[0]
    assert err < 1e-12 and active == virtual and msgs == 4 * (virtual - 1)     # 4 causal convolutions in _decode


def _subgroup_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from easyanimate_amd import vae_parallel
        # (1) partitions of the WORLD: created by every rank in the same order, cached (the second call returns the same objects)
        pairs = [[0, 1], [2, 3]]
        g1 = vae_parallel._subgroups(pairs, None)
        g2 = vae_parallel._subgroups(pairs, None)
        t = torch.tensor([float(rank)])
        dist.all_reduce(t, group=g1)
        ok_world = g1 is g2 and dist.get_world_size(g1) == 2 and t.item() == (1.0 if rank < 2 else 5.0)
        # (2) a partition of a STRICT sub-group (one CFG half): only its members get here; member-local creation
        ok_sub = True
        if rank < 2:
            singles = vae_parallel._subgroups([[0], [1]], g1)
            ok_sub = dist.get_world_size(singles) == 1 and vae_parallel._subgroups([[0], [1]], g1) is singles
        dist.barrier()
        ret[rank] = (ok_world, ok_sub)
    finally:
        dist.destroy_process_group()


def test_subgroups_are_cached_and_world_collective():
    """vae_parallel._subgroups (the CFG halves of sequence_parallel, the row groups of the VAE split): one creation per process,
    world-collective for partitions of the world (an eagerly initialised RCCL world splits communicators with every rank taking
    part), member-local only for partitions of a strict sub-group."""
    import torch.multiprocessing as mp
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_subgroup_worker, args=(4, _free_port(), ret), nprocs=4, join=True)
    assert len(ret) == 4 and all(a and b for a, b in ret.values()), dict(ret)


def _rebuilt_world_worker(rank, world, port, ret):
    from easyanimate_amd import vae_parallel
    seen, ok = [], True
    for it in range(3):                 # init -> destroy -> init in ONE process: every default group is named "0" again
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port + it))
        dist.init_process_group("gloo", rank=rank, world_size=world)
        g = vae_parallel._subgroups([[0], [1]], None)
        ok = ok and vae_parallel._subgroups([[0], [1]], None) is g and g not in seen        # cached inside a world, never across worlds
        t = torch.tensor([float(rank + 1)])
        dist.all_reduce(t, group=g)     # a group of the destroyed world would raise or hang here
        ok = ok and t.item() == float(rank + 1) and dist.distributed_c10d._get_default_group().group_name == "0"
        seen.append(g)
        dist.barrier()
        dist.destroy_process_group()
    ret[rank] = ok


def test_subgroup_cache_does_not_survive_its_world():
    """c10d restarts its group counter when the world is destroyed, so a cache keyed by the default group's name hands
    a rebuilt world the sub-groups of the destroyed one.  The cache is tied to the default-group object (weak reference, identity)
    and to c10d's registry: each world gets fresh, working groups."""
    import torch.multiprocessing as mp
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_rebuilt_world_worker, args=(2, _free_port(), ret), nprocs=2, join=True)
    assert len(ret) == 2 and all(ret.values()), dict(ret)
