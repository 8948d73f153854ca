#!/usr/bin/env python3
"""Stress the row-panel kernels while another process keeps the GPU busy: repeat the launch, compare every result with the first one bit for bit.
usage: python tools/rp_stress.py [iters] [tag]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from synergize_motion_appearance_amd import ops  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
tag = sys.argv[2] if len(sys.argv) > 2 else ""
BF = torch.bfloat16
ops.GEMM16_RP_MIN_ROWS = 1024
cases = []
for (M, K, N, res) in ((98304, 256, 128, False), (98304, 128, 128, True), (24576, 256, 256, True), (24576, 256, 512, False), (393216, 128, 128, False)):
    x = torch.randn((M, K), device="cuda").to(BF).view(6, -1, 1, K)
    cv = ops.Conv(torch.randn((N, K), device="cuda") / K ** 0.5, torch.randn((N,), device="cuda"), 1, 1, K, N)
    r = torch.randn((M, N), device="cuda").to(BF).view(6, -1, 1, N) if res else None
    ref = ops.conv(x, cv, res=r).clone()
    cases.append((x, cv, r, ref, (M, K, N, res)))
torch.cuda.synchronize()
bad = 0
for it in range(iters):
    for x, cv, r, ref, what in cases:
        y = ops.conv(x, cv, res=r)
        if not torch.equal(y, ref):
            d = (y.float() - ref.float()).abs()
            bad += 1
            if bad <= 5:
                print(tag, "MISMATCH it", it, what, "n", int((d > 0).sum()), "max", float(d.max()), flush=True)
print(tag, "done", iters, "iterations,", bad, "mismatching launches")
