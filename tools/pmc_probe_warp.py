#!/usr/bin/env python3
"""rocprofv3 --pmc workload: the A7 warp at s=128 / s=256, B=30 frames, broadcast source, smooth flows."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from synergize_motion_appearance_amd import ops  # noqa: E402

B = 30
grid = torch.stack(torch.meshgrid(torch.linspace(-1, 1, 64), torch.linspace(-1, 1, 64), indexing="xy"), -1).cuda()
th = torch.eye(2, device="cuda")[None] + 0.08 * torch.randn(B, 2, 2, device="cuda")
flow = (torch.einsum("bij,hwj->bhwi", th, grid) + 0.05 * torch.randn(B, 1, 1, 2, device="cuda")).contiguous()
occ = torch.rand(B, 64, 64, device="cuda")
for C, s in ((128, 128), (64, 256)):
    feat = torch.randn(1, s, s, C, device="cuda")
    out = torch.empty(B, s, s, C, device="cuda")
    for _ in range(3):
        ops.warp(feat, flow, occ, out=out)
torch.cuda.synchronize()
print("pmc probe done")
