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
