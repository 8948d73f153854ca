#!/usr/bin/env python3
"""Micro-benchmark of the fused Winograd kernel on the B=60 (and B=1) shapes of the step, per epilogue variant.
usage: python tools/wino_bench.py [B] [ablate]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from synergize_motion_appearance_amd import ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 60
ABLATE = len(sys.argv) > 2 and sys.argv[2] == "ablate"      # timing-only builds of the wide kernel with one phase removed
SHAPES = [(64, 64, 256), (128, 128, 128), (128, 64, 256), (128, 128, 256), (256, 128, 64), (128, 128, 64), (256, 256, 32), (256, 512, 32), (512, 256, 32)]


def timed(fn, n=5):
    for _ in range(4):                                       # the first runs after an idle gap read ~4 % low (clock ramp)
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


if ABLATE:
    print("executed MFMA fraction (of 157.3 TF) of the wide kernel: full us (frac) | full, no transform, no U loads, no region staging, no barriers, no epilogue, 1+2+4, all off "
          "| the same at 1 block/CU")
print(f"B={B}  cin cout s : wide blocks us (executed MFMA fraction of 157.3 TF) | other block shapes   [GN loader + stats + residual, as in a ResBlock]")
for cin, cout, s in SHAPES:
    x = torch.randn((B, s, s, cin), device="cuda")
    cv = ops.Conv.from_torch(torch.randn((cout, cin, 3, 3), device="cuda") / (3 * cin ** 0.5), torch.randn(cout, device="cuda") * 0.1)
    out = torch.empty((B, s, s, cout), device="cuda")
    res = torch.randn((B, s, s, cout), device="cuda")
    ss = torch.rand((B, cin, 2), device="cuda")
    fl = 2.0 * B * s * s * cout * 9 * cin
    row = []
    ops.set_tuning("wino_wide", 1)
    for abl in (0,):
        ops.set_tuning("wino_ablate", abl)
        t = timed(lambda: ops.conv(x, cv, out=out, in_ss=ss, in_swish=True, res=res, want_stats=True))
        row.append(f"{1e3 * t:7.1f} ({fl * 4 / 9 / t / 1e9 / 157.3:.3f})")
    ops.set_tuning("wino_ablate", 0)
    if ABLATE:
        for wide in (1, 5):                                  # 5: the same kernel at one block per CU (no co-resident partner wave)
            ops.set_tuning("wino_wide", wide)
            row.append("| 1 block/CU:" if wide == 5 else "|")
            for abl in (0, 1, 2, 4, 8, 16, 7, 31):
                ops.set_tuning("wino_ablate", abl)
                t = timed(lambda: ops.conv(x, cv, out=out, in_ss=ss, in_swish=True, res=res, want_stats=True))
                row.append(f"{fl * 4 / 9 / t / 1e9 / 157.3:.3f}")
            ops.set_tuning("wino_ablate", 0)
        ops.set_tuning("wino_wide", 1)
        print(f"{cin:4d} {cout:4d} {s:4d} : " + " ".join(row))
        continue
    ops.set_tuning("wino_nw", 1)
    t = timed(lambda: ops.conv(x, cv, out=out, in_ss=ss, in_swish=True, res=res, want_stats=True))
    row.append(f"N=32 blocks: {1e3 * t:7.1f} ({fl * 4 / 9 / t / 1e9 / 157.3:.3f})")
    ops.set_tuning("wino_nw", 2)
    ops.set_tuning("wino_wide", 0)
    t = timed(lambda: ops.conv(x, cv, out=out, in_ss=ss, in_swish=True, res=res, want_stats=True))
    row.append(f"8-wave N=64 blocks: {1e3 * t:7.1f} ({fl * 4 / 9 / t / 1e9 / 157.3:.3f})")
    ops.set_tuning("wino_nw", -1)
    ops.set_tuning("wino_wide", 1)
    print(f"{cin:4d} {cout:4d} {s:4d} : " + "   ".join(row))
