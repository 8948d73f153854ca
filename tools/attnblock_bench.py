#!/usr/bin/env python3
"""The AttnBlock (engine_netg._Attn) on bf16 storage at B frames: fused core (smx_attnblock_bf16) against the three-launch form.
usage: python tools/attnblock_bench.py [B]"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from synergize_motion_appearance_amd import ops, engine_netg as E  # noqa: E402
from tools.attn_bench import timed  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 300
C_ = 256
P = {"a.norm.weight": torch.ones(C_), "a.norm.bias": torch.zeros(C_)}
for n in ("q", "k", "v", "proj_out"):
    P[f"a.{n}.weight"] = torch.randn(C_, C_, 1, 1) / math.sqrt(C_)
    P[f"a.{n}.bias"] = torch.zeros(C_)
P = {k: v.cuda() for k, v in P.items()}
blk = E._Attn(P, "a")
x = torch.randn((B, 32, 32, C_), device="cuda").to(torch.bfloat16)
for flag in (1, 0):
    E.ATTNBLOCK_FUSED16 = flag
    with ops.profile() as rec:
        blk(x)
    t = timed(lambda: blk(x))
    print(f"B {B} {'fused core' if flag else 'three launches'}: block {1e3 * t:8.1f} us | " + "  ".join(f"{n} {1e3 * ms:7.1f}" for n, _, ms in rec.rows))
