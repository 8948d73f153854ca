import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from synergize_motion_appearance_amd import ops
from synergize_motion_appearance_amd.synth import synth_input
for (B, S, shared, masked) in ((2, 512, True, False), (2, 1024, False, False), (2, 256, True, False), (2, 1024, False, True), (2, 512, False, False), (2, 512, True, True)):
    q = synth_input(f"q{B}", (B, 1024, 256)).cuda()
    kv = synth_input(f"kv{B}{S}", ((1 if shared else B), 1024, 512)).cuda()
    k, v = (kv[0, :, :256], kv[0, :, 256:]) if shared else (kv[..., :256], kv[..., 256:])
    mask = None
    if masked:
        mask = torch.zeros((B, S), dtype=torch.uint8, device="cuda"); mask[:, 5::9] = 1
    r = {}
    for knob in (0, 19, 18):
        old = ops.set_tuning("attn_bf3", knob)
        r[knob] = ops.attention(q, k, v, 8, 32, S, k_shared=shared, mask=mask).clone()
        torch.cuda.synchronize()
        ops.set_tuning("attn_bf3", old)
    d19, d18 = (r[19] - r[0]).abs(), (r[18] - r[0]).abs()
    bad = (d18 > 1e-3).nonzero()
    print(B, S, shared, masked, "x6", float(d19.max()), "p2", float(d18.max()), "bad", bad.shape[0], bad[:5].tolist(), flush=True)
