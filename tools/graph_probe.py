#!/usr/bin/env python3
"""hipGraph feasibility probe: capture one B=1 / B=2 step (keypoints -> dense motion -> generator) with
torch.cuda.graph and compare replay with eager launches (values and time).  Result on MI355X: bit-identical,
8.01 vs 8.02 ms per step at B=1 -- the step is GPU-bound even at one frame, so graphs are not used."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, yaml
from basicsr.archs import build_network
from synergize_motion_appearance_amd import ops
from synergize_motion_appearance_amd.synth import synth_state_dict, synth_clip
cfg = yaml.safe_load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "options/test.yml")))
net_g, me = build_network(cfg["network_g"]), build_network(cfg["network_motion_estimator"])
net_g.load_state_dict(synth_state_dict([(k, v.shape) for k, v in net_g.state_dict().items()]))
me.load_state_dict(synth_state_dict([(k, v.shape) for k, v in me.state_dict().items()]))
net_g, me = net_g.cuda().eval(), me.cuda().eval()
src, drv = synth_clip(4, seed=5)
s = src[None].cuda(); d = drv.cuda()
eg, em = net_g.engine(), me.engine()
kp_s = em.estimate_kp(s); cache = eg.encode_source(s); src64 = em.source_down(s)
def step(frame, kp_s=kp_s):
    kp_d = em.estimate_kp(frame)
    dm = em.dense_motion(src64, kp_d, kp_s)
    st = eg.forward(cache, dm["deformation"], dm["occlusion_nhwc"].view(-1, 64, 64), dm["heat_nhwc"], 1.0)
    return st["out"]
for B in (1, 2):
    fr = d[:B].contiguous()
    for _ in range(3): ref = step(fr)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): step(fr)
    torch.cuda.synchronize(); te = (time.perf_counter() - t0) / 20
    static = fr.clone()
    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(g, capture_error_mode="relaxed"):
            out = step(static)
        g.replay(); torch.cuda.synchronize()
        print("B", B, "replay == eager:", torch.equal(out, ref))
        static.copy_(d[1:1 + B]); g.replay(); torch.cuda.synchronize()
        print("   new input matches eager:", torch.equal(out, step(d[1:1 + B].contiguous())))
        t0 = time.perf_counter()
        for _ in range(20): g.replay()
        torch.cuda.synchronize(); tg = (time.perf_counter() - t0) / 20
        print(f"   eager {te*1e3:.2f} ms/step, graph {tg*1e3:.2f} ms/step")
    except Exception as e:
        print("capture failed:", type(e).__name__, str(e)[:300])
