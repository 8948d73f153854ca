#!/usr/bin/env python3
"""Micro-benchmark of the fused attention kernels on the step's shapes (B frames x 8 heads x 1024 queries).
usage: python tools/attn_bench.py [B]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from synergize_motion_appearance_amd import ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 60


def timed(fn, n=10):
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


H, N = 8, 1024
for dh, E in (((4, 32), (32, 256)) if __name__ == "__main__" else ()):
    for S, shared, masked in ((1024, False, True), (1024, False, False), (256, True, False), (512, True, False), (768, True, False), (1024, True, False)):
        q = torch.randn((B, N, E), device="cuda")
        kv = torch.randn(((1 if shared else B), 1024, 2 * E), device="cuda")
        kd, vd = (kv[0, :S, :E], kv[0, :S, E:]) if shared else (kv[:, :S, :E], kv[:, :S, E:])
        mask = (torch.rand((B, S), device="cuda") < 0.05).to(torch.uint8) if masked else None
        row = []
        for knob in ((1, 0) if dh == 4 else (1,)):
            ops.set_tuning("attn4_mfma", knob)
            t = timed(lambda: ops.attention(q, kd, vd, H, dh, S, k_shared=shared, mask=mask))
            fl = 4.0 * B * H * N * S * dh
            row.append(f"{'mfma4x4' if (dh == 4 and knob) else ('valu' if dh == 4 else 'mfma32')}: {1e3 * t:7.1f} us ({fl / t / 1e9:6.1f} TF, {B * H * N * S / t / 1e6:7.1f} G pairs/s)")
        ops.set_tuning("attn4_mfma", 1)
        print(f"d_head {dh:2d} S {S:4d} {'shared kv' if shared else 'per-frame kv'}{' masked' if masked else ''}: " + "   ".join(row))
