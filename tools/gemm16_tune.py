#!/usr/bin/env python3
"""Tile sweep of gemm_bf16_kernel on the 1x1 / patch GEMM shapes of the B=60 bf16 step.  usage: python tools/gemm16_tune.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from synergize_motion_appearance_amd import ops  # noqa: E402

BF = torch.bfloat16
# (M, N, K) as 1x1 convolutions over M pixels
SCALE = int(sys.argv[1]) if len(sys.argv) > 1 else 1        # 5: the B = 300 launches
SHAPES = [(61440, 256, 256), (3932160, 64, 128), (61440, 512, 256), (61440, 4096, 256), (983040, 192, 128), (983040, 192, 64),
          (61440, 32, 32), (983040, 128, 256), (61440, 2048, 256), (245760, 128, 32), (61440, 64, 288), (245760, 128, 256), (245760, 192, 128),
          (61440, 256, 4096), (61440, 256, 1024)]


def timed(fn, n=8):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


print("M N K : auto us | tile1 128x128/8w  tile2 128x64/4w  tile3 64x64/4w  tile4 128x32  tile5 64x128  tile6 256x64/8w  tile7 128x128/4w   (GB/s of the best)")
for M, N, K in SHAPES:
    M *= SCALE
    x = torch.randn((1, M // 256, 256, K), device="cuda").to(BF)
    cv = ops.Conv.from_torch(torch.randn((N, K, 1, 1), device="cuda") / K ** 0.5, torch.randn(N, device="cuda") * 0.1)
    out = torch.empty((1, M // 256, 256, N), device="cuda", dtype=BF)
    ts = [timed(lambda t=t: ops.conv(x, cv, out=out, tile=t)) for t in (0, 1, 2, 3, 4, 5, 6, 7)]
    best = min(ts[1:])
    print(f"{M:8d} {N:5d} {K:5d} : {1e3 * ts[0]:7.1f} | " + " ".join(f"{1e3 * t:7.1f}" for t in ts[1:]) + f"   best tile {ts.index(best)}  {2.0 * M * (N + K) / best / 1e6:6.0f} GB/s {2.0 * M * N * K / best / 1e9:5.0f} TF")
