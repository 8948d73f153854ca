#!/usr/bin/env python3
"""F(4x4,3x3) (csrc/winograd43.hip) against the F(2x2,3x3) wide kernel on the B=60 shapes that carry the fp32 step, in the
ResBlock form (GroupNorm loader + residual + statistics).  usage: python tools/wino43_bench.py [B]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from synergize_motion_appearance_amd import ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 60
SHAPES = [(64, 64, 256), (128, 128, 128), (128, 64, 256), (128, 128, 256), (256, 128, 64), (128, 128, 64), (256, 256, 32), (256, 512, 32), (512, 256, 32)]


def timed(fn, n=5):
    for _ in range(4):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


print(f"B={B}  cin cout s : F(2,3) us (executed frac of 157.3 TF) | F(4,3) us (executed frac) | speed-up | max |diff|")
for cin, cout, s in SHAPES:
    x = torch.randn((B, s, s, cin), device="cuda")
    cv = ops.Conv.from_torch(torch.randn((cout, cin, 3, 3), device="cuda") / (3 * cin ** 0.5), torch.randn(cout, device="cuda") * 0.1)
    out, out2 = torch.empty((B, s, s, cout), device="cuda"), torch.empty((B, s, s, cout), device="cuda")
    res = torch.randn((B, s, s, cout), device="cuda")
    ss = torch.rand((B, cin, 2), device="cuda")
    fl = 2.0 * B * s * s * cout * 9 * cin
    ops.WINO43 = 0
    t0 = timed(lambda: ops.conv(x, cv, out=out, in_ss=ss, in_swish=True, res=res, want_stats=True))
    ops.WINO43 = 1
    t1 = timed(lambda: ops.conv(x, cv, out=out2, in_ss=ss, in_swish=True, res=res, want_stats=True))
    print(f"{cin:4d} {cout:4d} {s:4d} : {1e3 * t0:8.1f} ({fl * 4 / 9 / t0 / 1e9 / 157.3:.3f}) | {1e3 * t1:8.1f} ({fl * 2.25 / 9 / t1 / 1e9 / 157.3:.3f}) | "
          f"{t0 / t1:5.2f}x | {float((out - out2).abs().max()):.2e}")
