#!/usr/bin/env python3
"""Tile sweep of smx_gemm_conv_f32 on the 1x1 convolution / Linear shapes of the per-frame path (algorithmic TFLOP/s per tile id).
usage: python tools/gemm_tune_1x1.py [B=300]      (each point warmed for 0.1 s: the clocks ramp over tens of ms)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from synergize_motion_appearance_amd import ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 300
# (H, Cin, Cout)
SH = [(32, 256, 256), (32, 256, 512), (32, 512, 256), (32, 256, 4096), (32, 256, 2048), (256, 128, 64), (256, 64, 128), (128, 256, 128), (128, 128, 192), (128, 64, 192),
      (64, 128, 128), (128, 128, 256)]
TILES = [5, 8, 10, 11, 2, 4, 1, 6, 7, 9, 12, 13]
print(f"B={B}  H Cin Cout | auto   " + " ".join(f"t{t:<5d}" for t in TILES))
for (H, Cin, Cout) in SH:
    x = torch.randn(B, H, H, Cin, device="cuda")
    cv = ops.Conv.from_torch(torch.randn(Cout, Cin, 1, 1, device="cuda") * 0.05, torch.randn(Cout, device="cuda"))
    y = torch.empty(B, H, H, Cout, device="cuda")
    fl = 2.0 * B * H * H * Cout * Cin
    row = []
    for t in [0] + TILES:
        t0 = time.time()
        while time.time() - t0 < 0.1:
            ops.conv(x, cv, out=y, tile=t)
            torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.conv(x, cv, out=y, tile=t)
        e1.record()
        torch.cuda.synchronize()
        row.append(fl / (e0.elapsed_time(e1) / 10 * 1e-3) / 1e12)
    best = max(range(1, len(row)), key=lambda i: row[i])
    print(f"{H:4d} {Cin:4d} {Cout:4d} | " + " ".join(f"{v:6.1f}" for v in row) + f"   best t{TILES[best - 1]} ({row[best] / row[0]:.2f}x auto)")
