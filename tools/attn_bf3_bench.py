"""A/B of the split-bf16 attention kernel (attn_bf3_kernel) against the fp32-MFMA kernel on the fp32 configuration's launches (d_head 32, 8 heads).
usage: python tools/attn_bf3_bench.py [out.txt]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from synergize_motion_appearance_amd import ops
from synergize_motion_appearance_amd.synth import synth_input


def timeit(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


lines = ["B    S     kind          fp32-MFMA ms  TF     x6 ms   TF(alg)  speed-up   P-2-level ms  speed-up   x6, 128-query blocks ms   f16x3 ms  speed-up   max|x6 - fp32|"]
for (B, S, kind) in ((60, 1024, "self"), (60, 1024, "self+mask"), (60, 1024, "shared ctx"), (60, 256, "shared ctx"), (300, 1024, "self"), (38, 1024, "self")):
    q = synth_input(f"q{B}", (B, 1024, 256)).cuda()
    shared = kind == "shared ctx"
    kv = synth_input(f"kv{B}{S}", ((1 if shared else B), S, 512)).cuda()
    k, v = (kv[0, :, :256], kv[0, :, 256:]) if shared else (kv[..., :256], kv[..., 256:])
    mask = None
    if "mask" in kind:
        mask = torch.zeros((B, S), dtype=torch.uint8, device="cuda"); mask[:, 900:] = 1
    f = lambda: ops.attention(q, k, v, 8, 32, S, k_shared=shared, mask=mask)
    res, t = {}, {}
    for knob in (0, 19, 18, 51, 20):
        old = ops.set_tuning("attn_bf3", knob)
        res[knob] = f().clone(); t[knob] = timeit(f)
        ops.set_tuning("attn_bf3", old)
    fl = 4.0 * B * 8 * 1024 * S * 32
    lines.append(f"{B:<4d} {S:<5d} {kind:<13s} {t[0]:8.3f}    {fl / t[0] / 1e9:6.1f} {t[19]:8.3f}  {fl / t[19] / 1e9:6.1f}   {t[0] / t[19]:5.2f}x   {t[18]:8.3f}      {t[0] / t[18]:5.2f}x    {t[51]:8.3f}                {t[20]:8.3f}   {t[0] / t[20]:5.2f}x    {float((res[19] - res[0]).abs().max()):.2e}")
txt = "\n".join(lines)
print(txt)
if len(sys.argv) > 1:
    open(sys.argv[1], "w").write(txt + "\n")

# the fused AttnBlock core (one head of d = 256, V transposed)
from synergize_motion_appearance_amd import lib as L_
lines2 = ["AttnBlock core (d = 256, 1024 tokens): B   fp32-MFMA ms  TF    f16x3 ms  speed-up   max|diff|"]
for B in (60, 300):
    qk = synth_input(f"abqk{B}", (B, 1024, 512)).cuda(); vt = synth_input(f"abv{B}", (B, 256, 1024)).cuda(); o = torch.empty((B, 1024, 256), device="cuda")
    f = lambda: L_.check(L_.load().smx_attnblock_f32(qk.data_ptr(), 512, 1024 * 512, qk.data_ptr() + 1024, 512, 1024 * 512, vt.data_ptr(), 1024, 256 * 1024,
                                                     o.data_ptr(), 256, 1024 * 256, B, 1024, 1024, 256, 0.0625, ops._stream()), "smx_attnblock_f32")
    r, tt = {}, {}
    for knob in (0, 20):
        old = ops.set_tuning("attn_bf3", knob)
        f(); r[knob] = o.clone(); tt[knob] = timeit(f)
        ops.set_tuning("attn_bf3", old)
    fl = 4.0 * B * 1024 * 1024 * 256
    lines2.append(f"{B:<4d} {tt[0]:8.3f}  {fl / tt[0] / 1e9:6.1f} {tt[20]:8.3f}   {tt[0] / tt[20]:5.2f}x   {float((r[20] - r[0]).abs().max()):.2e}")
print("\n".join(lines2))
if len(sys.argv) > 1:
    open(sys.argv[1], "a").write("\n".join(lines2) + "\n")

