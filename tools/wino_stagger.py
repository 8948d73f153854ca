#!/usr/bin/env python3
"""Phase stagger of the wide Winograd kernel (csrc/winograd.hip `stagger`): the second block of every CU starts `stagger` shader cycles late, once per
launch, so that the two co-resident blocks stop running in lockstep.  Sweep over the delay on the bench's B = 300 shapes.
usage: python tools/wino_stagger.py [B]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from synergize_motion_appearance_amd import ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 300
SHAPES = [(64, 64, 256), (128, 128, 128), (128, 128, 256), (256, 128, 64), (256, 256, 32), (512, 256, 32)]
DELAYS = [0, 10000, 20000, 30000, 40000, 50000, 65000, 80000, 0]


def timed(fn, n=6):
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


print(f"B={B}: executed MFMA fraction (of 157.3 TF) of the wide Winograd kernel by stagger (shader cycles): " + " ".join(str(d) for d in DELAYS))
for cin, cout, s in SHAPES:
    x = torch.randn((B, s, s, cin), device="cuda")
    cv = ops.Conv.from_torch(torch.randn((cout, cin, 3, 3), device="cuda") / (3 * cin ** 0.5), torch.randn(cout, device="cuda") * 0.1)
    out, res, ss = torch.empty((B, s, s, cout), device="cuda"), torch.randn((B, s, s, cout), device="cuda"), torch.rand((B, cin, 2), device="cuda")
    fl = 2.0 * B * s * s * cout * 9 * cin
    row, ref = [], None
    for d in DELAYS:
        ops.set_tuning("wino_stagger", d)
        t = timed(lambda: ops.conv(x, cv, out=out, in_ss=ss, in_swish=True, res=res, want_stats=True))
        row.append(f"{fl * 4 / 9 / t / 1e9 / 157.3:.3f}")
        if ref is None:
            ref = out.clone()
        else:
            assert torch.equal(out, ref), "the stagger must not change the result"
    ops.set_tuning("wino_stagger", 0)
    print(f"{cin:4d}->{cout:4d} @{s:3d}: " + " ".join(row), flush=True)
