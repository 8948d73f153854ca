#!/usr/bin/env python3
"""Same-box A/B of the fp32 3x3 convolution: fp32-MFMA Winograd (wide kernel) | bf16x6 split kernel | bf16x3, per layer shape of the step
(ResBlock form: GroupNorm + swish loader, residual, GroupNorm partials), with the error of each against an fp64 convolution.
usage: python tools/wino_bf3_bench.py [B] [plain]"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from synergize_motion_appearance_amd import ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 300
PLAIN = len(sys.argv) > 2 and sys.argv[2] == "plain"
# (cin, cout, spatial): the heaviest fp32 3x3 launches of the step first (profiles/r05_gemm_shapes.txt)
SHAPES = [(128, 128, 128), (64, 64, 256), (256, 128, 64), (128, 64, 256), (256, 256, 32), (512, 256, 32), (256, 512, 32), (128, 128, 256), (64, 128, 256), (128, 128, 64)]


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


print(f"B={B}  {'plain conv' if PLAIN else 'ResBlock form (GN+swish loader, residual, GN partials)'}")
print("cin cout  s : fp32-MFMA us (frac of 157.3 TF fp32 pipe) | f16x3 us (frac of 2500 TF 16-bit pipe, 3 products) speedup | bf16x6 us (6 products) speedup | bf16x3 us speedup | "
      "max|err| vs fp64 on a 2-image sample: fp32-MFMA, f16x3, bf16x6, bf16x3")
ops.WINO_BF3_MIN_BLOCKS = 1
for cin, cout, s in SHAPES:
    x = torch.randn((B, s, s, cin), device="cuda")
    cv = ops.Conv.from_torch(torch.randn((cout, cin, 3, 3), device="cuda") / (3 * cin ** 0.5), torch.randn(cout, device="cuda") * 0.1)
    out = torch.empty((B, s, s, cout), device="cuda")
    res = None if PLAIN else torch.randn((B, s, s, cout), device="cuda")
    ss = None if PLAIN else torch.stack([1 + 0.2 * torch.rand((B, cin), device="cuda"), 0.1 * torch.randn((B, cin), device="cuda")], -1).contiguous()
    fl = 2.0 * B * s * s * cout * 4 * cin              # executed Winograd-domain multiplies x 2
    call = lambda: ops.conv(x, cv, out=out, in_ss=ss, in_swish=not PLAIN, res=res, want_stats=not PLAIN)
    # fp64 reference of two images (same loader arithmetic up to fp32 exp / rcp: compare the plain form for the error column)
    xs = x[:2]
    ref = F.conv2d(xs.permute(0, 3, 1, 2).double(), cv.w.view(cout, 3, 3, cin).permute(0, 3, 1, 2).double(), cv.b.double(), padding=1).permute(0, 2, 3, 1)
    row, errs = [], []
    t0 = None
    for mode in (0, 4, 6, 3):
        ops.WINO_BF3, ops.WINO_F16 = (6, 2) if mode == 4 else (mode, 0)
        t = timed(call)
        e = float((ops.conv(xs, cv).double() - ref).abs().max())
        errs.append(f"{e:.2e}")
        if mode == 0:
            t0 = t
            row.append(f"{1e3 * t:8.1f} ({fl / t / 1e9 / 157.3:.3f})")
        else:
            row.append(f"{1e3 * t:8.1f} ({fl * (3 if mode == 4 else mode) / t / 1e9 / 2500:.3f}) {t0 / t:4.2f}x")
    ops.WINO_BF3, ops.WINO_F16 = 0, 0
    print(f"{cin:3d} {cout:4d} {s:3d} : " + " | ".join(row) + " | " + " ".join(errs), flush=True)
    del x, out, res
