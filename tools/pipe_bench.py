#!/usr/bin/env python3
"""host-to-host FramePipeline throughput by batch size, eager launches vs hipGraph replay.  usage: python tools/pipe_bench.py [f32|bf16]"""
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
