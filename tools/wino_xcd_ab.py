#!/usr/bin/env python3
"""Wide Winograd kernel: the output blocks of a spatial tile grouped on one XCD (knob wino_xcd = 1) against the grid's natural order (0); same process,
same buffers, outputs and GroupNorm partials must be bit-identical.   usage: python tools/wino_xcd_ab.py [B]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from synergize_motion_appearance_amd import ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 300


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
    return e0.elapsed_time(e1) / n


print(f"B={B}: executed MFMA fraction of 157.3 TF, natural order | grouped | natural | grouped; identical")
for cin, cout, s in [(64, 64, 256), (128, 128, 128), (128, 128, 256), (256, 128, 64), (128, 256, 64), (256, 256, 32), (256, 512, 32), (512, 256, 32)]:
    x = torch.randn((B, s, s, cin), device="cuda")
    cv = ops.Conv.from_torch(torch.randn((cout, cin, 3, 3), device="cuda") / (3 * cin ** 0.5), torch.randn(cout, device="cuda") * 0.1)
    out, res, ss = torch.empty((B, s, s, cout), device="cuda"), torch.randn((B, s, s, cout), device="cuda"), torch.rand((B, cin, 2), device="cuda")
    fl = 2.0 * B * s * s * cout * 9 * cin * 4 / 9
    run = lambda: ops.conv(x, cv, out=out, in_ss=ss, in_swish=True, res=res, want_stats=True)   # noqa: E731
    row, ys = [], []
    for k in (0, 1, 0, 1):
        ops.set_tuning("wino_xcd", k)
        y = run(); ys.append((y.clone(), y._gn_part.clone()))
        row.append(f"{fl / timed(run) / 1e9 / 157.3:.3f}")
    same = torch.equal(ys[0][0], ys[1][0]) and torch.equal(ys[0][1], ys[1][1])
    print(f"{cin:4d}->{cout:4d} @{s:3d}: " + " | ".join(row) + f" ; {same}", flush=True)
ops.set_tuning("wino_xcd", 1)
