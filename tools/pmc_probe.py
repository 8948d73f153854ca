#!/usr/bin/env python3
"""Small fixed workload for rocprofv3 --pmc passes: the A7 warp at the four scales (B=10,
features broadcast from one source) and three dominant conv shapes, 3 launches each."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from synergize_motion_appearance_amd import ops  # noqa: E402

B = 10
flow = (torch.rand(B, 64, 64, 2, device="cuda") * 2 - 1) * 0.9
occ = torch.rand(B, 64, 64, device="cuda")
for C, s in ((256, 32), (128, 64), (128, 128), (64, 256)):
    feat = torch.randn(1, s, s, C, device="cuda")
    out = torch.empty(B, s, s, C, device="cuda")
    for _ in range(3):
        ops.warp(feat, flow, occ, out=out)
for (H, Cin, Cout) in ((128, 128, 128), (256, 64, 64), (64, 256, 256)):
    x = torch.randn(B, H, H, Cin, device="cuda")
    cv = ops.Conv.from_torch(torch.randn(Cout, Cin, 3, 3, device="cuda") * 0.03, torch.randn(Cout, device="cuda"))
    y = torch.empty(B, H, H, Cout, device="cuda")
    for _ in range(3):
        ops.conv(x, cv, out=y)
torch.cuda.synchronize()
print("pmc probe done")
