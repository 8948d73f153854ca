"""A/B of the split-bf16 row-panel GEMM (csrc/gemm_rp_bf3.hip) against the fp32-MFMA row-panel kernel on the fp32 configuration's 1x1 layers.
usage: python tools/gemm_bf3_bench.py [out.txt]"""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch

from synergize_motion_appearance_amd import ops
from synergize_motion_appearance_amd.synth import synth_input

# (M, K, N, d2s) at batch 60: token Linears 60*32*32 rows; 1x1 convolutions at 64^2 / 128^2
SHAPES = [(60 * 1024, 256, 256, None), (60 * 1024, 256, 512, None), (60 * 1024, 256, 1024, None), (60 * 1024, 256, 4096, (8, 64)), (60 * 1024, 256, 2048, (4, 128)),
          (60 * 4096, 256, 128, None), (60 * 4096, 128, 256, None), (60 * 16384, 128, 256, None), (60 * 16384, 256, 128, None)]


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


lines = ["M        K    N     d2s        fp32-MFMA ms  TF    f16x3 ms  TF(x3)  speed-up   bf16x6 ms  speed-up  max|f16x3 - fp32|"]
for (M, K, N, d2s) in SHAPES:
    H = 32 if M % 1024 == 0 and M // 1024 <= 64 else (64 if M // 4096 <= 64 else 128)
    B = M // (H * H)
    x = synth_input(f"gx{M}{K}", (B, H, H, K)).cuda()
    cv = ops.Conv((synth_input(f"gw{K}{N}", (N, K)) / math.sqrt(K)).cuda().contiguous(), synth_input(f"gb{N}", (N,)).cuda(), 1, 1, K, N)
    out = ops.conv(x, cv, d2s=d2s)
    ops.GEMM_RP_BF3, ops.GEMM_RP_F16 = 1, 0
    tb = timeit(lambda: ops.conv(x, cv, out=out, d2s=d2s))
    ops.GEMM_RP_F16 = 1
    with ops.profile() as rec:
        y6 = ops.conv(x, cv, out=out, d2s=d2s)
    used = [r[1].get("bf3") for r in rec.rows]
    t6 = timeit(lambda: ops.conv(x, cv, out=out, d2s=d2s))
    y6 = y6.clone()
    ops.GEMM_RP_BF3 = 0
    t32 = timeit(lambda: ops.conv(x, cv, out=out, d2s=d2s))
    y32 = ops.conv(x, cv, d2s=d2s)
    fl = 2.0 * M * K * N
    lines.append(f"{M:<8d} {K:<4d} {N:<5d} {str(d2s):<10s} {t32:8.3f}     {fl / t32 / 1e9:6.1f} {t6:8.3f}  {3 * fl / t6 / 1e9:6.1f}  {t32 / t6:6.2f}x   {tb:8.3f}  {t32 / tb:6.2f}x   {float((y6 - y32).abs().max()):.2e}  {used}")
txt = "\n".join(lines)
print(txt)
if len(sys.argv) > 1:
    open(sys.argv[1], "w").write(txt + "\n")
