#!/usr/bin/env python3
"""d_head-4 attention on bf16 storage: the bf16 4x4x4 MFMA kernel (attn4_mfma = 1) against the fp32 4x4x1 kernel (2) and the VALU kernel (0).
usage: python tools/attn4_bf16_bench.py [B]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from synergize_motion_appearance_amd import ops  # noqa: E402
from tools.attn_bench import timed  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 300
H, N, dh, E = 8, 1024, 4, 32
BF = torch.bfloat16
for S, shared, masked in ((1024, False, True), (1024, False, False), (256, True, False), (512, True, False), (768, True, False), (1024, True, False)):
    q = torch.randn((B, N, E), device="cuda").to(BF)
    kv = torch.randn(((1 if shared else B), 1024, 2 * E), device="cuda").to(BF)
    kd, vd = (kv[0, :S, :E], kv[0, :S, E:]) if shared else (kv[:, :S, :E], kv[:, :S, E:])
    mask = (torch.rand((B, S), device="cuda") < 0.05).to(torch.uint8) if masked else None
    row = []
    for knob, nm in ((1, "bf16 4x4x4"), (2, "f32 4x4x1"), (0, "valu")):
        ops.set_tuning("attn4_mfma", knob)
        t = timed(lambda: ops.attention(q, kd, vd, H, dh, S, k_shared=shared, mask=mask))
        row.append(f"{nm}: {1e3 * t:7.1f} us ({B * H * N * S / t / 1e6:7.1f} G pairs/s)")
    ops.set_tuning("attn4_mfma", 1)
    print(f"B {B} S {S:4d} {'shared kv' if shared else 'per-frame kv'}{' masked' if masked else ''}: " + "   ".join(row))
