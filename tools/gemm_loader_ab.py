#!/usr/bin/env python3
"""gemm_conv: the buffer-load loader (gemm_loader = 1, MODE 2) against the float4 gather (0), same process, same buffers; results must be bit-identical.
usage: python tools/gemm_loader_ab.py [B]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from synergize_motion_appearance_amd import ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 60


def timed(fn, n=5):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


# (cin, cout, k, stride, pad, size): the step's non-Winograd layers with C_in % 32 == 0
CASES = [(256, 256, 1, 1, 0, 32), (256, 512, 1, 1, 0, 32), (64, 128, 1, 1, 0, 256), (128, 17, 7, 1, 3, 64), (128, 76, 7, 1, 3, 58), (64, 128, 3, 2, 1, 256), (128, 256, 3, 2, 1, 128),
         (256, 512, 3, 2, 1, 64), (512, 1024, 3, 1, 1, 8), (256, 4096, 1, 1, 0, 32), (32, 64, 3, 1, 1, 64), (96, 64, 5, 1, 2, 40)]
print(f"B={B}: cin cout k stride size | gather us | buffer loads us | speed-up | identical")
for cin, cout, k, st, pad, s in CASES:
    x = torch.randn((B, s, s, cin), device="cuda")
    cv = ops.Conv.from_torch(torch.randn((cout, cin, k, k), device="cuda") / (k * cin ** 0.5), torch.randn(cout, device="cuda") * 0.1)
    run = lambda: ops.conv(x, cv, stride=st, pad=(pad, pad), act=ops.ACT_LRELU02 if hasattr(ops, "ACT_LRELU02") else ops.ACT_NONE, direct=True)   # noqa: E731
    ops.set_tuning("gemm_loader", 0)
    y0 = run().clone(); t0 = timed(run)
    ops.set_tuning("gemm_loader", 1)
    y1 = run().clone(); t1 = timed(run)
    fl = 2.0 * y0.numel() * cin * k * k
    print(f"  {cin:4d} {cout:5d} {k} {st} {s:4d} | {t0:9.1f} ({fl / t0 / 1e6:6.1f} TF) | {t1:9.1f} ({fl / t1 / 1e6:6.1f} TF) | {t0 / t1:5.3f} | {bool(torch.equal(y0, y1))}", flush=True)
