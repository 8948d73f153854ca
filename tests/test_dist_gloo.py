"""The N>1 path on CPU: world_size-2 `gloo` processes exercise exactly the host logic the GPU
run uses over RCCL -- one broadcast of the packed source cache from rank 0, contiguous frame
shards with no data-path collective, ordered gather of the per-rank results."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_frames, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from synergize_motion_appearance_amd import driver
    from synergize_motion_appearance_amd.engine_netg import SourceCache
    from synergize_motion_appearance_amd.synth import synth_input
    flat = None
    if rank == 0:       # only the root owns the source encoding
        feats = {s: synth_input(f"cache{s}", sh) for s, sh in driver.CACHE_SHAPES.items()}
        kp = {"value": synth_input("cv", (1, 15, 2)), "jacobian": synth_input("cj", (1, 15, 2, 2))}
        flat = driver.pack_source_state(SourceCache(feats, 1), synth_input("s64", (1, 64, 64, 3)), kp, kp, 1.5)
    buf = driver.broadcast_flat(flat, torch.device("cpu"), src=0)
    st = driver.unpack_source_state(buf)
    assert st.scale == 1.5
    cache, kp_s = st.cache, st.kp_source
    # every frame is independent given the cache: a rank's "render" depends on (cache, frame id) only
    a, b = driver.shard_frames(n_frames, rank, world)
    mine = torch.tensor([float(cache.feats[32].sum() + kp_s["value"].sum()) + t for t in range(a, b)], dtype=torch.float64)
    sizes = [driver.shard_frames(n_frames, r, world)[1] - driver.shard_frames(n_frames, r, world)[0] for r in range(world)]
    parts = [torch.empty(n, dtype=torch.float64) for n in sizes]
    dist.all_gather(parts, mine) if len(set(sizes)) == 1 else None
    if len(set(sizes)) != 1:                              # ragged shards: gather with padding
        pad = torch.zeros(max(sizes), dtype=torch.float64)
        pad[:mine.numel()] = mine
        allp = [torch.empty(max(sizes), dtype=torch.float64) for _ in range(world)]
        dist.all_gather(allp, pad)
        parts = [p[:n] for p, n in zip(allp, sizes)]
    q.put((rank, float(buf.double().sum()), torch.cat(parts).tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_world2_broadcast_and_frame_sharding():
    world, n_frames = 2, 301
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_frames, q)) for r in range(world)]
    [p.start() for p in procs]
    res = sorted(q.get(timeout=120) for _ in range(world))
    [p.join(timeout=60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    (r0, sum0, frames0), (r1, sum1, frames1) = res
    assert sum0 == sum1                                    # both ranks hold the identical broadcast cache
    assert frames0 == frames1 and len(frames0) == n_frames
    base = frames0[0]
    assert [round(f - base) for f in frames0] == list(range(n_frames))   # contiguous shards, in order, none lost


# ---- animate_sharded's host logic with stub engines (no GPU): strict subgroup + empty shard ------------------------------
class _StubEngine:
    """stands in for BOTH engines: every output is a simple function of (source mean, frame mean), so the gathered clip
    can be checked exactly; the collectives, rank conventions and shard arithmetic are the real driver code."""
    adt = torch.float32

    def estimate_kp(self, x):
        B = x.shape[0]
        return {"value": x.mean((1, 2, 3)).view(B, 1, 1).expand(B, 15, 2).contiguous(),
                "jacobian": torch.eye(2).expand(B, 15, 2, 2).contiguous()}

    def encode_source(self, src):
        from synergize_motion_appearance_amd import driver
        from synergize_motion_appearance_amd.engine_netg import SourceCache
        return SourceCache({s: torch.full(sh, float(src.mean())) for s, sh in driver.CACHE_SHAPES.items()}, 1)

    def source_down(self, src):
        return torch.zeros(1, 64, 64, 3)

    def dense_motion(self, src64, kp_n, kp_s):
        B = kp_n["value"].shape[0]
        return {"deformation": kp_n["value"][:, 0, 0].view(B, 1, 1, 1).expand(B, 64, 64, 2).contiguous(),
                "occlusion_nhwc": torch.zeros(B, 64, 64, 1), "heat_nhwc": torch.zeros(B, 64, 64, 15)}

    def forward(self, cache, deformation, occ, heat, w):
        B = deformation.shape[0]
        v = cache.feats[32].reshape(-1)[0] + deformation[:, 0, 0, 0]
        return {"out": v.view(B, 1, 1, 1).expand(B, 8, 8, 3).contiguous()}


class _StubNet:
    def engine(self):
        return _StubEngine()

    def parameters(self):
        return iter([torch.zeros(1)])


def _subgroup_worker(rank, world, port, members, n_frames, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from synergize_motion_appearance_amd import driver
    driver.ops.to_uint8 = lambda x, lo, hi: ((x.clamp(lo, hi) + 1.0) * 127.5).round().to(torch.uint8)   # CPU stand-in for the packer
    group = dist.new_group(members)                      # every rank takes part in creating it
    res = None
    if rank in members:
        gen = torch.Generator().manual_seed(5)
        source = torch.rand(3, 8, 8, generator=gen) * 0.2 - 0.5
        driving = (torch.arange(n_frames, dtype=torch.float32) * 0.01).view(-1, 1, 1, 1).expand(n_frames, 3, 8, 8).contiguous()
        if dist.get_rank(group) != 0:                    # only the group's root holds the source
            source = torch.full_like(source, float("nan"))
        out = driver.animate_sharded(source, driving, _StubNet(), _StubNet(), relative=False, adapt_movement_scale=False,
                                     batch=2, root=0, gather=True, group=group)
        span, mine = driver.animate_sharded(source, driving, _StubNet(), _StubNet(), relative=False, adapt_movement_scale=False,
                                            batch=2, root=0, gather=False, group=group)
        res = (rank, None if out is None else out[:, 0, 0, 0].tolist(), span, tuple(mine.shape))
    q.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()


def _run_subgroup(world, members, n_frames):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_subgroup_worker, args=(r, world, port, members, n_frames, q)) for r in range(world)]
    [p.start() for p in procs]
    res = dict(q.get(timeout=120) for _ in range(world))
    [p.join(timeout=60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    return res


def _expected(n_frames):
    gen = torch.Generator().manual_seed(5)
    src_mean = float((torch.rand(3, 8, 8, generator=gen) * 0.2 - 0.5).mean())
    v = torch.tensor([src_mean + 0.01 * t for t in range(n_frames)], dtype=torch.float32)
    return ((v.clamp(-1, 1) + 1.0) * 127.5).round().to(torch.uint8).tolist()


def test_animate_sharded_in_a_strict_subgroup_uses_group_local_root():
    """`root` is a rank of `group`; global ranks 1-2 of a 3-process job with root=0 -> global rank 1 encodes,
    broadcasts and gathers (rank 0 of the job is not a member and takes no part)."""
    res = _run_subgroup(3, [1, 2], 7)
    assert res[0] is None
    assert res[1][1] == _expected(7) and res[2][1] is None           # the group's root (global 1) holds the clip, in order
    assert res[1][2] == (0, 4) and res[2][2] == (4, 7)
    assert res[1][3] == (4, 8, 8, 3) and res[2][3] == (3, 8, 8, 3)


def test_animate_sharded_with_an_empty_shard_does_not_hang():
    """fewer frames than ranks -> one rank's shard is empty; it must still reach the gather."""
    res = _run_subgroup(2, [0, 1], 1)
    assert res[0][1] == _expected(1) and res[1][1] is None
    assert res[0][2] == (0, 1) and res[1][2] == (1, 1) and res[1][3] == (0, 8, 8, 3)


# ---- the training step's gradient collective (DDP as bucketed all-reduces of one flat buffer), host tensors -------------
def _flat_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from synergize_motion_appearance_amd.trainer import FlatParams
    torch.manual_seed(rank)                                       # replicas that start apart (BasicSR seeds rank r with seed + r)
    net = torch.nn.Sequential(torch.nn.Linear(37, 101), torch.nn.Linear(101, 5))
    flat = FlatParams(net, allow_cpu=True)
    flat.m.fill_(float(rank)), flat.v.fill_(2.0 * rank)
    flat.t = 7 * (1 - rank)
    before = flat.value.clone()
    flat.broadcast(dist, src=0)                                   # DDP's construction-time broadcast (+ the Adam state, for a resume)
    synced = (rank == 0 and torch.equal(before, flat.value)) or (rank == 1 and not torch.equal(before, flat.value))
    synced = synced and float(flat.m.abs().sum()) == 0.0 and float(flat.v.abs().sum()) == 0.0 and flat.t == 7
    synced = synced and all(p.data_ptr() == flat.P[n].data_ptr() for n, p in net.named_parameters())
    # parameters alias the flat buffer (checkpoint names / shapes intact), slots are 256-B aligned
    assert all(p.data_ptr() == flat.P[n].data_ptr() for n, p in net.named_parameters())
    assert all(o % FlatParams.ALIGN == 0 for o, _ in flat.slots.values())
    for n, g in flat.G.items():
        g.fill_(float(rank + 1))                                  # rank r contributes r+1 everywhere
    works = flat.all_reduce_start(dist, bucket_mb=0.001)          # 262-float buckets: many collectives in flight, joined later
    other = torch.ones(5) * rank                                  # "the other network's backward" runs meanwhile
    flat.all_reduce_wait(works)
    ok = all(bool((g == 3.0).all()) for g in flat.G.values()) and len(works) > 4 and float(other.sum()) == 5.0 * rank     # 1 + 2
    pad_untouched = float(flat.grad.sum()) == 3.0 * sum(n for _, n in flat.slots.values())
    q.put((rank, ok and synced, pad_untouched, [tuple(p.grad.shape) for p in net.parameters()], float(flat.value.double().sum())))
    dist.barrier()
    dist.destroy_process_group()


def test_world2_flat_gradient_all_reduce():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_flat_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in procs]
    res = sorted(q.get(timeout=120) for _ in range(world))
    [p.join(timeout=60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    for rank, ok, pad, shapes, chk in res:
        assert ok and pad, rank
        assert shapes == [(101, 37), (101,), (5, 101), (5,)]
    assert res[0][4] == res[1][4]                                 # identical replicas after the broadcast
