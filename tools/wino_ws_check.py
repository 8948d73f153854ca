#!/usr/bin/env python3
"""The producer / consumer-split Winograd kernel (csrc/winograd.hip winograd_ws_kernel, smx_set_tuning("wino_ws", 1)) against the wide kernel:
bit-identical outputs and GroupNorm partials over the epilogue / loader forms, then timing at the bench's B = 300 shapes.
usage: python tools/wino_ws_check.py [B]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from synergize_motion_appearance_amd import ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 300
g = torch.Generator().manual_seed(1)
rnd = lambda *s: torch.randn(*s, generator=g).cuda()   # noqa: E731
ops.set_tuning("wino_nw", 2)
bad = 0
for (b, cin, cout, h, w, gn, res, act, up2, stats) in [(4, 64, 64, 32, 32, 2, True, 0, False, True), (2, 128, 128, 16, 64, 1, False, 1, False, False), (3, 128, 64, 24, 32, 0, True, 3, False, True),
                                                      (2, 256, 128, 16, 32, 2, True, 0, True, True), (1, 64, 192, 64, 64, 0, False, 2, False, False), (5, 32, 64, 8, 64, 2, True, 4, False, True),
                                                      (8, 128, 128, 64, 64, 2, True, 0, False, True)]:
    hs, ws_ = (h // 2, w // 2) if up2 else (h, w)
    x = rnd(b, hs, ws_, cin)
    cv = ops.Conv.from_torch(rnd(cout, cin, 3, 3) / (3 * cin ** 0.5), rnd(cout) * 0.1)
    r = rnd(b, h, w, cout) if res else None
    ss = torch.rand((b, cin, 2), generator=g).cuda() if gn else None
    outs = []
    for wsf in (0, 1):
        ops.set_tuning("wino_ws", wsf)
        y = ops.conv(x, cv, res=r, in_ss=ss, in_swish=(gn == 2), act=act, up2=up2, want_stats=stats)
        torch.cuda.synchronize()
        outs.append((y.clone(), getattr(y, "_gn_part", None).clone() if stats and getattr(y, "_gn_part", None) is not None else None))
    same = torch.equal(outs[0][0], outs[1][0]) and (outs[0][1] is None or torch.equal(outs[0][1], outs[1][1]))
    d = float((outs[0][0] - outs[1][0]).abs().max())
    if outs[0][1] is not None and not torch.equal(outs[0][1], outs[1][1]):
        dp = (outs[0][1] - outs[1][1]).abs()
        idx = (dp > 0).nonzero()
        print("   partials differ:", int((dp > 0).sum()), "of", dp.numel(), "max", float(dp.max()), "first", idx[:3].tolist(),
              [float(outs[0][1][tuple(i)]) for i in idx[:3]], [float(outs[1][1][tuple(i)]) for i in idx[:3]], flush=True)
    print(f"B={b} {cin}->{cout} @{h}x{w} gn={gn} res={res} act={act} up2={up2} stats={stats}: {'identical' if same else 'DIFFERENT max ' + str(d)}", flush=True)
    bad += not same
ops.set_tuning("wino_nw", -1)
print("correctness:", "OK" if not bad else f"{bad} cases differ", flush=True)


def timed(fn, n=5):
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


print(f"B={B}: executed MFMA fraction of 157.3 TF, wide kernel | split kernel | wide again")
for cin, cout, s in [(64, 64, 256), (128, 128, 128), (128, 128, 256), (256, 128, 64), (256, 256, 32), (512, 256, 32)]:
    x = torch.randn((B, s, s, cin), device="cuda")
    cv = ops.Conv.from_torch(torch.randn((cout, cin, 3, 3), device="cuda") / (3 * cin ** 0.5), torch.randn(cout, device="cuda") * 0.1)
    out, res, ss = torch.empty((B, s, s, cout), device="cuda"), torch.randn((B, s, s, cout), device="cuda"), torch.rand((B, cin, 2), device="cuda")
    fl = 2.0 * B * s * s * cout * 9 * cin
    row = []
    for wsf in (0, 1, 0):
        ops.set_tuning("wino_ws", wsf)
        t = timed(lambda: ops.conv(x, cv, out=out, in_ss=ss, in_swish=True, res=res, want_stats=True))
        row.append(f"{fl * 4 / 9 / t / 1e9 / 157.3:.3f}")
    ops.set_tuning("wino_ws", 0)
    print(f"{cin:4d}->{cout:4d} @{s:3d}: " + " | ".join(row), flush=True)
