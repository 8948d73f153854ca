#!/usr/bin/env python3
"""One product kernel in a loop, as the OTHER process next to tools/sm_probe: which kernels of the bf16 pipeline make a
co-resident process's waves go wrong?   usage: python tools/aggressor.py <which> <seconds>
  rp16       row-panel GEMM, bf16 (LDS-DMA A tiles, persistent blocks)          t32       3x3 conv, 16x32 tiles (LDS-DMA weights + region)
  t32gn      the same with the GroupNorm+swish loader (region through registers) conv16    3x3 region kernel, 16x16 tiles (no LDS-DMA)
  gemm16     implicit GEMM, bf16 (no LDS-DMA)                                      c7x3      7x7 heads, bf16x3 (LDS-DMA)
  attnblock  fused AttnBlock (LDS-DMA K / V^T tiles)                               attn32    d_head-32 attention, bf16 MFMA (no LDS-DMA)
  attn4      d_head-4 attention, bf16 4x4x4 MFMA                                   smalln    C_out <= 4 3x3 on the bf16 MFMA (LDS-DMA)
  wino       fp32 Winograd (no LDS-DMA)                                            rp32      row-panel GEMM, fp32 (LDS-DMA)
"""
import math
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from synergize_motion_appearance_amd import ops  # noqa: E402

which, secs = sys.argv[1], float(sys.argv[2])
BF = torch.bfloat16
dev = "cuda"
g = torch.Generator(device="cpu").manual_seed(5)
rnd = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dev)   # noqa: E731
ops.GEMM16_RP_MIN_ROWS = 1024
ops.CONV16_T32_MIN_BLOCKS = 1
ops.SMALLN_MFMA_MIN_BLOCKS = 1
B = 6
if which in ("rp16", "gemm16", "rp32"):
    K, N = 256, 256
    x = rnd(B, 64, 64, K)
    x = x.to(BF) if which != "rp32" else x
    cv = ops.Conv(rnd(N, K, sc=1 / 16).contiguous(), rnd(N), 1, 1, K, N)
    if which == "gemm16":
        ops.GEMM16_RP = 0
    fn = lambda: ops.conv(x, cv)   # noqa: E731
elif which in ("t32", "t32gn", "conv16", "wino"):
    Cin = Cout = 128
    x = rnd(B, 64, 64, Cin)
    x = x.to(BF) if which != "wino" else x
    cv = ops.Conv.from_torch(rnd(Cout, Cin, 3, 3, sc=1 / math.sqrt(9 * Cin)), rnd(Cout))
    ss = None
    if which == "t32gn":
        ss = ops.groupnorm_stats(x, torch.ones(Cin, device=dev), torch.zeros(Cin, device=dev))
    if which == "conv16":
        ops.CONV16_T32 = 0
    fn = lambda: ops.conv(x, cv, in_ss=ss, in_swish=ss is not None)   # noqa: E731
elif which.startswith("t32abl"):
    # "t32abl:<mask>[:gn]": the tools build of the 16x32-tile kernel with phases compiled out (tools/conv_t32_ablate.py --build -> tools/conv_t32_tools.bin):
    # 1 no region loads, 2 no weight DMA, 4 no region store, 8 no MFMA, 16 no fragment reads, 32 no epilogue traffic, 64 no epilogue
    import ctypes as C
    parts = which.split(":")
    abl, gn = int(parts[1]), int(len(parts) > 2)
    lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "conv_t32_tools.bin"))
    _p, _i = C.c_void_p, C.c_int
    lib.smx_conv3x3_bf16_t32_abl.argtypes = [_p, _i, _p, _p, _p, _i, _i, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p, _i, _p, _p, _i]
    Cin = Cout = 128
    x = rnd(B, 64, 64, Cin).to(BF)
    cv = ops.Conv.from_torch(rnd(Cout, Cin, 3, 3, sc=1 / math.sqrt(9 * Cin)), rnd(Cout))
    wp, out, ss = cv.w16_t32, torch.empty((B, 64, 64, Cout), device=dev, dtype=BF), torch.rand((B, Cin, 2), device=dev)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def fn():
        rc = lib.smx_conv3x3_bf16_t32_abl(x.data_ptr(), Cin, wp.data_ptr(), cv.b.data_ptr(), None, 0, 0, out.data_ptr(), Cout, B, 64, 64, Cin, Cout, 0, 0,
                                          ss.data_ptr() if gn else None, gn, None, st, abl)
        assert rc == 0, rc
elif which == "smallf":
    x = rnd(B, 64, 64, 128)
    cv = ops.Conv.from_torch(rnd(2, 128, 3, 3, sc=1 / math.sqrt(9 * 128)), rnd(2))
    fn = lambda: ops.conv(x, cv)   # noqa: E731
elif which in ("warp", "warp16"):
    feat = rnd(1, 64, 64, 128)
    feat = feat.to(BF) if which == "warp16" else feat
    flow = (torch.rand((B, 64, 64, 2), generator=g) * 2 - 1).to(dev)
    occ = torch.rand((B, 64, 64), generator=g).to(dev)
    fn = lambda: ops.warp(feat, flow, occ)   # noqa: E731
elif which in ("gn", "gn16"):
    x = rnd(B, 64, 64, 128)
    x = x.to(BF) if which == "gn16" else x
    gam, bet = torch.ones(128, device=dev), torch.zeros(128, device=dev)
    fn = lambda: ops.groupnorm(x, gam, bet, swish=True)   # noqa: E731
elif which == "c7x3":
    x = rnd(B, 64, 64, 128)
    cv = ops.Conv.from_torch(rnd(17, 128, 7, 7, sc=1 / math.sqrt(49 * 128)), rnd(17))
    fn = lambda: ops.conv7_x3(x, cv, pad=3)   # noqa: E731
elif which == "attnblock":
    from synergize_motion_appearance_amd import engine_netg as E
    C_ = 256
    P = {"a.norm.weight": torch.ones(C_, device=dev), "a.norm.bias": torch.zeros(C_, device=dev)}
    for n in ("q", "k", "v", "proj_out"):
        P[f"a.{n}.weight"], P[f"a.{n}.bias"] = rnd(C_, C_, 1, 1, sc=1.5 / 16), rnd(C_, sc=0.1)
    blk = E._Attn(P, "a")
    x = rnd(B, 32, 32, C_).to(BF)
    fn = lambda: blk(x)   # noqa: E731
elif which in ("attn32", "attn4"):
    dh, H = (32, 8) if which == "attn32" else (4, 8)
    E_ = dh * H
    q, k, v = rnd(B, 1024, E_).to(BF), rnd(B, 1024, E_).to(BF), rnd(B, 1024, E_).to(BF)
    fn = lambda: ops.attention(q, k, v, H, dh, 1024)   # noqa: E731
elif which == "smalln":
    x = rnd(B, 64, 64, 128).to(BF)
    cv = ops.Conv.from_torch(rnd(2, 128, 3, 3, sc=1 / math.sqrt(9 * 128)), rnd(2))
    fn = lambda: ops.conv(x, cv, out_dtype=torch.float32)   # noqa: E731
else:
    raise SystemExit(__doc__)
with ops.profile() as rec:
    fn()
print(f"[aggressor {which}] launches per call: {[r[0] + (':rp' if (r[1] or {}).get('rp') else '') + (':t32' if (r[1] or {}).get('t32') else '') for r in rec.rows]}", flush=True)
t_end, n = time.time() + secs, 0
if len(sys.argv) > 3 and sys.argv[3] == "check":
    # victim mode: every output is compared with the first one, bit for bit (is THIS kernel hurt by what runs next to it?)
    ref = fn().clone()
    bad = worst = 0
    while time.time() < t_end:
        flags = []
        for _ in range(20):
            y = fn()
            flags.append((y != ref).sum())
        f = torch.stack(flags).cpu()
        bad += int((f > 0).sum())
        worst = max(worst, int(f.max()))
        n += 20
    print(f"[victim {which}] {n} calls, {bad} with a wrong output (worst: {worst} elements)", flush=True)
    sys.exit(0)
while time.time() < t_end:
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    n += 50
print(f"[aggressor {which}] {n} calls", flush=True)
