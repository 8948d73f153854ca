#!/usr/bin/env python3
"""time the A7 warp kernel at the four scales (B frames, source features broadcast) -> algorithmic GB/s
usage: warp_bench.py [B] [noise|smooth]   (SMX_TOOLS build + SMX_WARP_ROWS=0: the per-lane-coordinates kernel for comparison)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from synergize_motion_appearance_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 20
flow = (torch.rand(B, 64, 64, 2, device="cuda") * 2 - 1) * 0.95
grid = torch.stack(torch.meshgrid(torch.linspace(-1, 1, 64), torch.linspace(-1, 1, 64), indexing="xy"), -1).cuda()
mode = sys.argv[2] if len(sys.argv) > 2 else "noise"
if mode == "noise":      # rough: per-cell noise of 0.1 (12.8 px at s=256)
    flow = grid[None] + 0.1 * torch.randn(B, 64, 64, 2, device="cuda")
else:                    # smooth: per-frame affine (what the motion estimator produces on talking heads)
    th = torch.eye(2, device="cuda")[None] + 0.08 * torch.randn(B, 2, 2, device="cuda")
    flow = (torch.einsum("bij,hwj->bhwi", th, grid) + 0.05 * torch.randn(B, 1, 1, 2, device="cuda")).contiguous()
occ = torch.rand(B, 64, 64, device="cuda")
tot_b = tot_t = 0
for C, s in ((256, 32), (128, 64), (128, 128), (64, 256)):
    feat = torch.randn(1, s, s, C, device="cuda")
    out = torch.empty(B, s, s, C, device="cuda")
    run = lambda: ops.warp(feat, flow, occ, out=out)  # noqa: E731
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n):
        run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    by = 4.0 * (2 * B * s * s * C + B * 64 * 64 * 3)
    tot_b += by; tot_t += ms
    print(f"s={s:3d} C={C:3d}: {ms*1e3:7.1f} us  {by/ms/1e6:7.1f} GB/s algorithmic")
print(f"all four: {tot_b/tot_t/1e6:7.1f} GB/s")
