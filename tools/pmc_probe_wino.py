#!/usr/bin/env python3
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from synergize_motion_appearance_amd import ops
B = 10
for (H, Cin, Cout) in ((128, 128, 128), (256, 64, 64), (64, 256, 256), (32, 256, 512)):
    x = torch.randn(B, H, H, Cin, device="cuda")
    cv = ops.Conv.from_torch(torch.randn(Cout, Cin, 3, 3, device="cuda") * 0.03, torch.randn(Cout, device="cuda"))
    y = torch.empty(B, H, H, Cout, device="cuda")
    for _ in range(3):
        ops.conv(x, cv, out=y)
torch.cuda.synchronize()
print("probe done")
