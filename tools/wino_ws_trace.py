#!/usr/bin/env python3
"""Where an MFMA wave of the split Winograd kernel spends its life (SMX_TOOLS build, wino_ablate = 32: s_memtime totals per MFMA wave
written over the GroupNorm partials), and the shader clock it ran at (s_memtime against the 100 MHz s_memrealtime).
usage: SMX_TOOLS=1 python tools/wino_ws_trace.py [B]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from synergize_motion_appearance_amd import ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 300
NAMES = ["restart (U requests, first slot, first transform)", "sub-step loop", "  of which: waiting for a ring slot", "drain (vmcnt 0)", "wait for the exchange buffer", "hand-over (Z -> LDS, clear, signal)"]
HNAMES = ["stage: addresses + loads issued", "stage: wait for a free slot", "stage: data wait, GN+swish, LDS store", "epilogue: residual loads issued", "epilogue: wait for the accumulators", "epilogue: work"]
ops.set_tuning("wino_ws", 1)
ops.set_tuning("wino_ablate", 32)
for cin, cout, s in [(64, 64, 256), (128, 128, 128), (256, 128, 64), (512, 256, 32)]:
    x = torch.randn((B, s, s, cin), device="cuda")
    cv = ops.Conv.from_torch(torch.randn((cout, cin, 3, 3), device="cuda") / (3 * cin ** 0.5), torch.randn(cout, device="cuda") * 0.1)
    res, ss = torch.randn((B, s, s, cout), device="cuda"), torch.rand((B, cin, 2), device="cuda")
    for _ in range(3):
        out = ops.conv(x, cv, in_ss=ss, in_swish=True, res=res, want_stats=True)
    torch.cuda.synchronize()
    raw = out._gn_part.view(torch.int32).flatten()[:256 * 5 * 16].cpu().numpy().astype(np.int64) & 0xFFFFFFFF
    tr = raw[:256 * 4 * 16].reshape(256, 4, 16).astype(np.float64)
    hp = raw[256 * 4 * 16:].reshape(256, 16).astype(np.float64)
    life, real, tiles = tr[:, :, 0], tr[:, :, 1], tr[:, :, 9]
    nsl = cin // 32
    floor = nsl * 128 * 64
    print(f"{cin}->{cout} @ {s}: {tiles.mean():.1f} tiles per block, wave life {life.mean():.0f} cycles = {life.mean() / tiles.mean():.0f} per tile (MFMA floor {floor}: "
          f"{floor * tiles.mean() / life.mean():.3f}); shader clock {life.sum() / real.sum() * 100:.0f} MHz")
    for k, n in enumerate(NAMES):
        v = tr[:, :, 2 + k] / tiles
        print(f"    {n:55s} {v.mean():9.0f} cycles per tile  {100 * tr[:, :, 2 + k].sum() / life.sum():5.1f} %")
    for k, n in enumerate(HNAMES):
        print(f"    helper wave 4: {n:40s} {(hp[:, k] / tiles[:, 0]).mean():9.0f} cycles per tile  {100 * hp[:, k].sum() / life[:, 0].sum():5.1f} %")
ops.set_tuning("wino_ablate", 0)
ops.set_tuning("wino_ws", 0)
