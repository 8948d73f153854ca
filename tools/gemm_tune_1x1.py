#!/usr/bin/env python3
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from synergize_motion_appearance_amd import ops
SH = [(60, 32, 256, 256), (60, 32, 256, 512), (60, 32, 256, 4096), (60, 32, 256, 2048), (60, 256, 64, 128), (60, 128, 128, 192), (60, 64, 128, 128), (60, 128, 128, 256)]
TILES = [5, 8, 10, 11, 2, 4, 1, 6, 7, 9, 12]
print("B H Cin Cout | " + " ".join(f"t{t:<5d}" for t in TILES))
for (B, H, Cin, Cout) in SH:
    x = torch.randn(B, H, H, Cin, device="cuda")
    cv = ops.Conv.from_torch(torch.randn(Cout, Cin, 1, 1, device="cuda") * 0.05, torch.randn(Cout, device="cuda"))
    y = torch.empty(B, H, H, Cout, device="cuda")
    fl = 2.0 * B * H * H * Cout * Cin
    row = []
    for t in TILES:
        for _ in range(2): ops.conv(x, cv, out=y, tile=t)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): ops.conv(x, cv, out=y, tile=t)
        e1.record(); torch.cuda.synchronize()
        row.append(fl / (e0.elapsed_time(e1) / 10 * 1e-3) / 1e12)
    print(f"{B} {H} {Cin} {Cout} | " + " ".join(f"{v:6.1f}" for v in row))
