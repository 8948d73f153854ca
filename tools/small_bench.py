#!/usr/bin/env python3
"""time the C_out<=4 3x3 conv (VALU kernel vs matrix-core path) at the two shapes of the pipeline"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from synergize_motion_appearance_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 30
for C, s, gn in ((64, 256, True), (256, 64, False)):
    x = torch.randn(B, s, s, C, device="cuda")
    cv = ops.Conv.from_torch(torch.randn(3, C, 3, 3, device="cuda") * 0.05, torch.randn(3, device="cuda"))
    ss = ops.groupnorm_stats(x, torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")) if gn else None
    for small in (True, False):
        ops.SMALLN = small
        for _ in range(3):
            ops.conv(x, cv, in_ss=ss)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.conv(x, cv, in_ss=ss)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print(f"C={C:3d} s={s:3d} gn={int(gn)} {'VALU small-N' if small else 'matrix cores '}: {ms*1e3:7.1f} us  {x.numel()*4/ms/1e6:7.1f} GB/s read")
