import os, sys, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from synergize_motion_appearance_amd import ops
ops.WINO_BF3_MIN_BLOCKS = 1
torch.manual_seed(0)
def run(x, w, mode, pad=0):
    ops.WINO_BF3 = mode
    cv = ops.Conv.from_torch(w.cuda(), None)
    xin = x.permute(0, 2, 3, 1).contiguous().cuda()
    if pad:
        big = torch.zeros(xin.shape[:3] + (xin.shape[3] + pad,), device="cuda"); big[..., pad:] = xin; xin = big[..., pad:]
    return ops.conv(xin, cv).permute(0, 3, 1, 2).cpu()
for (B, C, Co, H, pad) in ((1, 128, 64, 32, 0), (1, 128, 64, 32, 32), (1, 96, 64, 32, 0), (1, 128, 64, 16, 0), (2, 128, 128, 64, 0), (1, 256, 64, 32, 0)):
    x = torch.randn(B, C, H, H); w = torch.randn(Co, C, 3, 3) / (3 * C ** 0.5)
    ref = F.conv2d(x.double(), w.double(), padding=1).float()
    y = run(x, w, 6, pad); y0 = run(x, w, 0, pad)
    e = (y - ref).abs()
    print(f"B{B} C{C} Co{Co} H{H} pad{pad}: fp32 {float((y0-ref).abs().max()):.2e} bf3 {float(e.max()):.2e}  by y%4 {[round(float(e[:, :, i::4].max()), 2) for i in range(4)]} by x%4 {[round(float(e[..., i::4].max()), 2) for i in range(4)]}")
    # per-slice impulse: x nonzero only in one 32-channel slice
    for sl in range(C // 32):
        xs = torch.zeros_like(x); xs[:, sl * 32:(sl + 1) * 32] = x[:, sl * 32:(sl + 1) * 32]
        r = F.conv2d(xs.double(), w.double(), padding=1).float()
        ys = run(xs, w, 6, pad)
        es = (ys - r).abs()
        print(f"    slice {sl}: err {float(es.max()):.2e}  odd rows {float(es[:, :, 1::2].max()):.2e} even rows {float(es[:, :, 0::2].max()):.2e}  ch<32 {float(es[:, :32].max()):.2e} ch>=32 {float(es[:, 32:].max()):.2e}")
