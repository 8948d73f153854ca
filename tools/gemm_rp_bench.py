#!/usr/bin/env python3
"""Short-K 1x1 layers on bf16 storage: the row-panel kernel (csrc/gemm_rp_bf16.hip) against the implicit GEMM, at the step's shapes.
usage: python tools/gemm_rp_bench.py [B] [f32]   (f32: the fp32 kernel, csrc/gemm_rp_f32.hip)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from synergize_motion_appearance_amd import ops  # noqa: E402
from tools.attn_bench import timed  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 300
F32 = len(sys.argv) > 2 and sys.argv[2] == "f32"
BF = torch.float32 if F32 else torch.bfloat16
# (pixels per frame, K, N, residual)
for px, K, N, res in ((1024, 256, 256, True), (1024, 256, 256, False), (1024, 256, 512, False), (1024, 256, 4096, False), (1024, 256, 2048, False),
                      (16384, 256, 128, False), (4096, 256, 128, False), (65536, 128, 128, False)):
    M = B * px
    x = torch.randn((M, K), device="cuda").to(BF).view(B, -1, 1, K)
    cv = ops.Conv(torch.randn((N, K), device="cuda") / K ** 0.5, torch.randn((N,), device="cuda"), 1, 1, K, N)
    r = torch.randn((M, N), device="cuda").to(BF).view(B, -1, 1, N) if res else None
    out = torch.empty((B, px, 1, N), device="cuda", dtype=BF)
    row = []
    for flag in (1, 0):
        ops.GEMM16_RP = ops.GEMM_RP = flag
        t = timed(lambda: ops.conv(x, cv, out=out, res=r))
        by = (4.0 if F32 else 2.0) * M * (K + N * (2 if res else 1))
        row.append(f"{'row-panel' if flag else 'implicit GEMM'}: {1e3 * t:7.1f} us ({2.0 * M * N * K / t / 1e9:6.1f} TF, {by / t / 1e6:6.0f} GB/s)")
    ops.GEMM16_RP = ops.GEMM_RP = 1
    print(f"M {M:9d} K {K} N {N:4d}{' +res' if res else '     '}: " + "   ".join(row))
