import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn.functional as F
from synergize_motion_appearance_amd import ops
from synergize_motion_appearance_amd.synth import synth_input
ops.WINO_BF3_MIN_BLOCKS = 1
def t(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return 1e3 * e0.elapsed_time(e1) / n
for (B, C, Co, H) in ((2, 128, 128, 64), (2, 64, 64, 64), (2, 256, 128, 32), (1, 512, 256, 32)):
    x = synth_input(f"x{C}{H}", (B, C, H, H)) * 1.5 + 0.2
    g, bt = 1 + 0.1 * synth_input(f"g{C}", (C,)), 0.1 * synth_input(f"b{C}", (C,))
    w = synth_input(f"w{C}{Co}", (Co, C, 3, 3)) / math.sqrt(9 * C); b = synth_input(f"bb{Co}", (Co,)) * 0.1
    ref = F.conv2d(F.silu(F.group_norm(x.double(), 32, g.double(), bt.double(), 1e-6)), w.double(), b.double(), padding=1)
    xin = x.permute(0, 2, 3, 1).contiguous().cuda()
    ss = ops.groupnorm_stats(xin, g.cuda(), bt.cuda())
    cv = ops.Conv.from_torch(w.cuda(), b.cuda())
    out = {}
    for name, bf3, f16 in (("fp32", 0, 0), ("x6", 6, 0), ("f16x3", 6, 1)):
        ops.WINO_BF3, ops.WINO_F16 = bf3, f16
        with ops.profile() as rec:
            y = ops.conv(xin, cv, in_ss=ss, in_swish=True)
        out[name] = (y.permute(0, 3, 1, 2).cpu().double() - ref)
        print(C, Co, H, name, rec.rows[0][1].get("bf3"), "max %.3e rms %.3e" % (float(out[name].abs().max()), float(out[name].pow(2).mean().sqrt())), flush=True)
# raw inputs (no GroupNorm): plain, tiny-valued, and a tensor whose later channels outgrow the first slice by 1e5 (the rescale path)
for (name_, B, C, Co, H, mk) in (("plain", 2, 128, 128, 32, lambda x: x), ("tiny 1e-4", 2, 128, 64, 32, lambda x: x * 1e-4), ("huge 3e3", 1, 64, 64, 32, lambda x: x * 3e3),
                                 ("grow 1e5", 2, 128, 128, 32, lambda x: torch.cat([x[:, :32] * 1e-3, x[:, 32:] * 100.0], 1)),
                                 ("grow late", 2, 256, 128, 32, lambda x: torch.cat([x[:, :200], x[:, 200:] * 3e4], 1)), ("zeros first", 1, 64, 64, 32, lambda x: torch.cat([x[:, :32] * 0, x[:, 32:]], 1))):
    x = mk(synth_input(f"rx{C}{H}{name_}", (B, C, H, H)))
    w = synth_input(f"rw{C}{Co}", (Co, C, 3, 3)) / math.sqrt(9 * C); b = synth_input(f"rb{Co}", (Co,)) * 0.1
    r = synth_input(f"rr{Co}{H}", (B, Co, H, H))
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1) + r.double()
    xin = x.permute(0, 2, 3, 1).contiguous().cuda(); rin = r.permute(0, 2, 3, 1).contiguous().cuda()
    cv = ops.Conv.from_torch(w.cuda(), b.cuda())
    for name, bf3, f16 in (("fp32", 0, 0), ("x6", 6, 0), ("f16x3", 6, 2)):
        ops.WINO_BF3, ops.WINO_F16 = bf3, f16
        with ops.profile() as rec:
            y = ops.conv(xin, cv, res=rin)
        d = (y.permute(0, 3, 1, 2).cpu().double() - ref)
        print(name_, name, rec.rows[0][1].get("bf3"), "max %.3e rms %.3e  (ref rms %.2e)" % (float(d.abs().max()), float(d.pow(2).mean().sqrt()), float(ref.pow(2).mean().sqrt())), flush=True)
# timing at B = 300, ResBlock form
for (C, Co, S) in ((128, 128, 128), (64, 64, 256), (256, 128, 64), (512, 256, 32)):
    Bt = 300
    x = torch.randn((Bt, S, S, C), device="cuda"); res = torch.randn((Bt, S, S, Co), device="cuda"); outb = torch.empty((Bt, S, S, Co), device="cuda")
    cv = ops.Conv.from_torch(torch.randn((Co, C, 3, 3), device="cuda") / (3 * C ** 0.5), torch.randn(Co, device="cuda") * 0.1)
    ss = torch.stack([1 + 0.2 * torch.rand((Bt, C), device="cuda"), 0.1 * torch.randn((Bt, C), device="cuda")], -1).contiguous()
    row = []
    for name, bf3, f16 in (("fp32", 0, 0), ("x6", 6, 0), ("f16x3", 6, 1)):
        ops.WINO_BF3, ops.WINO_F16 = bf3, f16
        row.append("%s %8.1f" % (name, t(lambda: ops.conv(x, cv, out=outb, in_ss=ss, in_swish=True, res=res, want_stats=True))))
    for name, bf3, f16 in (("plain x6", 6, 0), ("plain f16x3", 6, 2)):
        ops.WINO_BF3, ops.WINO_F16 = bf3, f16
        row.append("%s %8.1f" % (name, t(lambda: ops.conv(x, cv, out=outb, res=res))))
    print(f"{C}->{Co}@{S}: " + " | ".join(row), flush=True)
    del x, res, outb
