#!/usr/bin/env python3
"""Weight-gradient GEMM (csrc/train_gemm.hip wgrad_kernel, fp32 and bf16-MFMA forms) on the conv shapes of the training step at B pairs.
usage: python tools/wgrad_bench.py [B]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from synergize_motion_appearance_amd import lib as L  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
lib = L.load()
# (cin, cout, s, k)
SHAPES = [(64, 64, 256, 3), (128, 128, 256, 3), (128, 128, 128, 3), (128, 64, 256, 3), (256, 256, 32, 3), (256, 512, 32, 3), (512, 256, 32, 3), (256, 128, 64, 3), (128, 128, 64, 3),
          (256, 256, 32, 1), (256, 512, 32, 1), (1024, 512, 8, 3), (128, 256, 32, 3), (35, 76, 64, 7)]
st = torch.cuda.current_stream().cuda_stream
print(f"B={B}: cin cout s k : generic fp32 us (TF) | bf16-MFMA us (TF) | region kernel (wgrad_region=1; 3x3 shapes only) fp32 | bf16-MFMA | msplit")
for cin, cout, s, k in SHAPES:
    M = B * s * s
    x = torch.randn(B, s, s, cin, device="cuda")
    dy = torch.randn(B, s, s, cout, device="cuda")
    out = torch.zeros(cout, cin, k, k, device="cuda")
    bias = torch.zeros(cout, device="cuda")
    ms = C.c_int(1)
    res = []
    for fn, knob in ((lib.smx_wgrad_f32, 0), (lib.smx_wgrad_mfma16_f32, 0), (lib.smx_wgrad_f32, 1), (lib.smx_wgrad_mfma16_f32, 1)):
        lib.smx_set_tuning(b"wgrad_region", knob)
        n = int(lib.smx_wgrad_conv_ws_floats(1, M, cout, cin, s, s, s, s, k, k, 1, k // 2, k // 2, 0, C.byref(ms)))
        ws = torch.empty(n, device="cuda")
        def run():
            L.check(fn(dy.data_ptr(), cout, 0, x.data_ptr(), cin, 0, 1, M, cout, s, s, cin, s, s, k, k, 1, k // 2, k // 2, 0, ws.data_ptr(), ms.value,
                       out.data_ptr(), 0, 0, 0, 0, 1.0, bias.data_ptr(), st), "wgrad")
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            run()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100
        res.append((us, 2.0 * M * cout * k * k * cin / us / 1e6))
    print(f"  {cin:4d} {cout:4d} {s:3d} {k} : {res[0][0]:8.1f} ({res[0][1]:5.1f}) | {res[1][0]:8.1f} ({res[1][1]:6.1f}) | region: {res[2][0]:8.1f} ({res[2][1]:5.1f}) | {res[3][0]:8.1f} ({res[3][1]:6.1f}) | {ms.value}")
