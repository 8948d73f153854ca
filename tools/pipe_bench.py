#!/usr/bin/env python3
"""host-to-host FramePipeline throughput by batch size, eager launches vs hipGraph replay.  usage: python tools/pipe_bench.py [f32|bf16]
       python tools/pipe_bench.py [f32|bf16] --from-png --to-png [frames] [batch]: folder of PNG frames -> basicsr.demo.animate_folder -> folder of PNG frames (frames/s)"""
import os
import sys
import time

import torch
import yaml

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from basicsr.archs import build_network  # noqa: E402
from synergize_motion_appearance_amd import driver, ops  # noqa: E402
from synergize_motion_appearance_amd.synth import synth_state_dict, synth_clip  # noqa: E402

dt = sys.argv[1] if len(sys.argv) > 1 else "f32"
cfg = yaml.safe_load(open(os.path.join(REPO, "options/test.yml")))
net_g, me = build_network(cfg["network_g"]), build_network(cfg["network_motion_estimator"])
net_g.load_state_dict(synth_state_dict([(k, v.shape) for k, v in net_g.state_dict().items()]))
me.load_state_dict(synth_state_dict([(k, v.shape) for k, v in me.state_dict().items()]))
net_g, me = net_g.cuda().eval(), me.cuda().eval()
net_g.set_compute_dtype(dt)
me.set_compute_dtype(dt)
if "--from-png" in sys.argv:
    import shutil
    import tempfile
    from basicsr.demo import animate_folder
    from synergize_motion_appearance_amd.png import encode_many, default_workers
    nums = [int(a) for a in sys.argv[2:] if a.isdigit()]
    n, B = (nums + [300, 60])[:2]
    src, drv = synth_clip(n, seed=1)
    u8 = ops.to_uint8(drv.cuda().permute(0, 2, 3, 1).contiguous(), -1.0, 1.0).cpu().numpy()
    s8 = ops.to_uint8(src[None].cuda().permute(0, 2, 3, 1).contiguous(), -1.0, 1.0)[0].cpu().numpy()
    root = tempfile.mkdtemp(prefix="smx_pipe_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        din, dout = os.path.join(root, "in"), os.path.join(root, "out")
        os.makedirs(din)
        encode_many(list(u8), [os.path.join(din, f"{i:06d}.png") for i in range(n)], level=1)
        pipe = driver.FramePipeline(net_g, me, batch=B, frame_hw=(256, 256))
        animate_folder(net_g, me, s8, din, dout, True, True, 0, B, pipe=pipe)          # warm: graph capture, pools
        ts = []
        for _ in range(3):
            shutil.rmtree(dout)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            animate_folder(net_g, me, s8, din, dout, True, True, 0, B, pipe=pipe)
            ts.append(time.perf_counter() - t0)
        print(f"{dt}: folder -> folder, {n} frames, batch {B}, {default_workers()} codec threads: {n / min(ts):.1f} frames/s (best of 3; {[round(n / t, 1) for t in ts]})")
    finally:
        shutil.rmtree(root, ignore_errors=True)
    sys.exit(0)
src, drv = synth_clip(120, seed=1)
u8 = ops.to_uint8(drv.cuda().permute(0, 2, 3, 1).contiguous(), -1.0, 1.0).cpu().pin_memory()
st = driver.encode_source_state(net_g, me, src.cuda(), drv[0:1].cuda(), True)
print(f"{dt}: batch : eager fps | graph fps")
for B in (1, 4, 8, 16, 30, 60):
    res = []
    for ug in (False, True):
        pipe = driver.FramePipeline(net_g, me, batch=B, use_graph=ug)
        n = (120 // B) * B
        out = torch.empty((n, 256, 256, 3), dtype=torch.uint8).pin_memory()
        pipe.run(st, u8[:n], out)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(2):
            pipe.run(st, u8[:n], out)
        torch.cuda.synchronize()
        res.append(2 * n / (time.perf_counter() - t0))
    print(f"  {B:3d} : {res[0]:8.1f} | {res[1]:8.1f}")
