#!/usr/bin/env python3
"""Micro-benchmark of the bf16 3x3 convolution kernels on the B=60 shapes of the step (region-direct vs implicit GEMM).
usage: python tools/conv16_bench.py [B]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from synergize_motion_appearance_amd import ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 60
SHAPES = [(64, 64, 256), (128, 128, 128), (128, 64, 256), (128, 128, 256), (256, 128, 64), (128, 128, 64), (256, 256, 32), (256, 512, 32), (512, 256, 32), (256, 128, 128)]
BF = torch.bfloat16


def timed(fn, n=6):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


print(f"B={B}  cin cout s : region us (TF, GB/s in+out)   implicit-GEMM us (TF)   +GN loader us   +stats us")
for cin, cout, s in SHAPES:
    x = torch.randn((B, s, s, cin), device="cuda").to(BF)
    cv = ops.Conv.from_torch(torch.randn((cout, cin, 3, 3), device="cuda") / (3 * cin ** 0.5), torch.randn(cout, device="cuda") * 0.1)
    out = torch.empty((B, s, s, cout), device="cuda", dtype=BF)
    ss = torch.rand((B, cin, 2), device="cuda")
    fl = 2.0 * B * s * s * cout * 9 * cin
    by = 2.0 * B * s * s * (cin + cout)
    ops.CONV16_TILE_H = 16
    ops.CONV16_T32 = 1
    t_t = timed(lambda: ops.conv(x, cv, out=out))
    t_tn = timed(lambda: ops.conv(x, cv, out=out, in_ss=ss, in_swish=True))
    t_ts = timed(lambda: ops.conv(x, cv, out=out, want_stats=True))
    res = torch.randn((B, s, s, cout), device="cuda").to(BF)
    t_tr = timed(lambda: ops.conv(x, cv, out=out, res=res, in_ss=ss, in_swish=True, want_stats=True))
    ops.CONV16_T32 = 0
    t_or = timed(lambda: ops.conv(x, cv, out=out, res=res, in_ss=ss, in_swish=True, want_stats=True))
    print(f"{cin:4d} {cout:4d} {s:4d} : t32 {1e3 * t_t:8.1f} us ({fl / t_t / 1e9:6.0f} TF, {by / t_t / 1e6:6.0f} GB/s)  +GN {1e3 * t_tn:8.1f} ({fl / t_tn / 1e9:6.0f} TF)  +stats {1e3 * t_ts:8.1f}  "
          f"GN+res+stats {1e3 * t_tr:8.1f} ({fl / t_tr / 1e9:6.0f} TF)  [16x16 form: {1e3 * t_or:8.1f} ({fl / t_or / 1e9:6.0f} TF)]")
    t_r = timed(lambda: ops.conv(x, cv, out=out))
    ops.CONV16_TILE_H = 8
    t_8 = timed(lambda: ops.conv(x, cv, out=out))
    t_8n = timed(lambda: ops.conv(x, cv, out=out, in_ss=ss, in_swish=True))
    ops.CONV16_TILE_H = 16
    t_g = timed(lambda: ops.conv(x, cv, out=out, tile=2 if cout <= 64 else 1))
    t_n = timed(lambda: ops.conv(x, cv, out=out, in_ss=ss, in_swish=True))
    t_s = timed(lambda: ops.conv(x, cv, out=out, want_stats=True))
    print(f"{cin:4d} {cout:4d} {s:4d} : {1e3 * t_r:8.1f} ({fl / t_r / 1e9:6.0f} TF, {by / t_r / 1e6:6.0f} GB/s)   {1e3 * t_g:8.1f} ({fl / t_g / 1e9:6.0f} TF)   "
          f"{1e3 * t_n:8.1f}   {1e3 * t_s:8.1f}   | tile_h=8: {1e3 * t_8:8.1f} ({fl / t_8 / 1e9:5.0f} TF)  +GN {1e3 * t_8n:8.1f}")
