#!/usr/bin/env python3
"""a few launches of the split-bf16 Winograd kernel (ResBlock form) for a rocprofv3 --pmc pass: tools/pmc_kernel.sh <tag> "<counters>" python tools/pmc_probe_bf3.py [B] [mode]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from synergize_motion_appearance_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 60
ops.WINO_BF3 = int(sys.argv[2]) if len(sys.argv) > 2 else 6
ops.WINO_BF3_MIN_BLOCKS = 1
for (H, Cin, Cout) in ((128, 128, 128),):
    x = torch.randn(B, H, H, Cin, device="cuda")
    cv = ops.Conv.from_torch(torch.randn(Cout, Cin, 3, 3, device="cuda") * 0.03, torch.randn(Cout, device="cuda"))
    y = torch.empty(B, H, H, Cout, device="cuda"); res = torch.randn(B, H, H, Cout, device="cuda")
    ss = torch.stack([1 + 0.2 * torch.rand((B, Cin), device="cuda"), 0.1 * torch.randn((B, Cin), device="cuda")], -1).contiguous()
    for _ in range(3):
        ops.conv(x, cv, out=y, in_ss=ss, in_swish=True, res=res, want_stats=True)
torch.cuda.synchronize()
print("probe done")
