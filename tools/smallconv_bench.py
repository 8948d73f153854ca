#!/usr/bin/env python3
"""3x3 convolutions with few channels at B=60 (motion-branch layers): fused Winograd (4-wave block) vs implicit GEMM tiles."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from synergize_motion_appearance_amd import ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 60
SHAPES = [(32, 32, 32), (32, 64, 32), (64, 32, 32), (128, 96, 64), (160, 126, 64), (128, 64, 64), (192, 128, 64), (32, 32, 64), (256, 128, 64)]


def timed(fn, n=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


print(f"B={B} cin cout s : winograd us (alg TF) | gemm tile5 64x64 | tile3 128x32 | tile15 64x32 | tile10 128x64/8w")
for cin, cout, s in SHAPES:
    x = torch.randn((B, s, s, cin), device="cuda")
    cv = ops.Conv.from_torch(torch.randn((cout, cin, 3, 3), device="cuda") / (3 * cin ** 0.5), torch.randn(cout, device="cuda") * 0.1)
    out = torch.empty((B, s, s, cout), device="cuda")
    fl = 2.0 * B * s * s * cout * 9 * cin
    ts = [timed(lambda t=t: ops.conv(x, cv, out=out, tile=t)) for t in (0, 5, 3, 15, 10)]
    print(f"{cin:4d} {cout:4d} {s:4d} : " + " | ".join(f"{1e3 * t:7.1f} ({fl / t / 1e9:4.0f})" for t in ts))
