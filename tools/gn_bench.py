#!/usr/bin/env python3
"""time groupnorm_stats (partial + finalize) at the decoder shapes -> read GB/s"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from synergize_motion_appearance_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 30
for C, s in ((256, 32), (256, 64), (128, 64), (128, 128), (64, 256), (128, 256)):
    x = torch.randn(B, s, s, C, device="cuda")
    g, b = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
    for _ in range(3):
        ops.groupnorm_stats(x, g, b)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.groupnorm_stats(x, g, b)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"s={s:3d} C={C:3d}: {ms*1e3:7.1f} us  {x.numel()*4/ms/1e6:7.1f} GB/s read")
