#!/usr/bin/env python3
"""Per-wave phase timeline of the wide Winograd kernel (timing-only build of the kernel, wino_ablate = 32: s_memtime stamps
written over the GroupNorm partials).  Prints the mean cycles a wave spends in each phase of a block.
usage: python tools/wino_trace.py [B] [cin] [cout] [s]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from synergize_motion_appearance_amd import ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 60
cin = int(sys.argv[2]) if len(sys.argv) > 2 else 128
cout = int(sys.argv[3]) if len(sys.argv) > 3 else 128
s = int(sys.argv[4]) if len(sys.argv) > 4 else 256

x = torch.randn((B, s, s, cin), device="cuda")
cv = ops.Conv.from_torch(torch.randn((cout, cin, 3, 3), device="cuda") / (3 * cin ** 0.5), torch.randn(cout, device="cuda") * 0.1)
res = torch.randn((B, s, s, cout), device="cuda")
ss = torch.rand((B, cin, 2), device="cuda")
ops.set_tuning("wino_wide", 1)
ops.set_tuning("wino_ablate", 32)
for _ in range(2):
    out = ops.conv(x, cv, in_ss=ss, in_swish=True, res=res, want_stats=True)
torch.cuda.synchronize()
part = out._gn_part                                        # [B, chunks, cout, 2] floats = per (chunk, nblk): 4 waves x 32 u32
raw = part.view(torch.int32).cpu().numpy().astype(np.int64) & 0xFFFFFFFF
tr = raw.reshape(-1, cout // 64, 4, 32)                    # [chunk, nblk, wave, stamp]
tr = tr.reshape(-1, 4, 32)
nsl = cin // 32
t0 = tr[:, :, 0]


def d(a, b):
    return ((tr[:, :, b] - tr[:, :, a]) & 0xFFFFFFFF).astype(np.float64)


rows = [("prologue: entry -> first region's loads issued", d(0, 24)), ("prologue: wait, GN+swish, LDS store", d(24, 25)), ("prologue: U prefetch issued", d(25, 1)), ("prologue barrier", d(1, 2))]
base = 2
for k in range(min(nsl, 4)):
    a, b, c = 3 + 3 * k, 4 + 3 * k, 5 + 3 * k
    rows.append((f"slice {k}: 4 x (transform + 32 MFMA)", d(base, a)))
    rows.append((f"slice {k}: stage next region (GN+swish, LDS store)", d(a, b)))
    rows.append((f"slice {k}: barrier", d(b, c)))
    base = c
if nsl > 4:
    rows.append((f"slices 4..{nsl - 1}", d(base, 18) * 0))
rows += [("(slice 1 sub 0 / 1 / 2 issue time)", None), ("epilogue: first barrier (from last slice barrier)", d(base, 18)),
         ("epilogue: accumulators -> LDS", d(18, 19)), ("epilogue: barrier", d(19, 20)),
         ("epilogue: LDS -> inverse transform, residual, stores, stats values", d(20, 21)), ("epilogue: stats reduction", d(21, 22)),
         ("whole block (wave lifetime)", d(0, 22))]
tot = d(0, 22).mean()
print(f"B={B} {cin}->{cout} @ {s}x{s}: {tr.shape[0]} blocks, mean wave lifetime {tot:.0f} cycles; MFMA issue floor {nsl * 128 * 64} cycles/wave "
      f"({nsl * 128 * 64 / tot:.3f} of lifetime; two waves share a SIMD)")
for name, v in rows:
    if v is None:
        s1 = [d(5, 15).mean(), d(15, 16).mean(), d(16, 17).mean(), d(17, 6).mean()]
        print(f"  {'slice 1 per sub (transform + 32 MFMA issue)':62s} " + " / ".join(f"{q:6.0f}" for q in s1))
        continue
    print(f"  {name:62s} {v.mean():8.0f} cyc  {100 * v.mean() / tot:5.1f} %   (p10 {np.percentile(v, 10):7.0f}  p90 {np.percentile(v, 90):7.0f})")
ops.set_tuning("wino_ablate", 0)
