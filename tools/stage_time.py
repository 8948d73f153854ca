#!/usr/bin/env python3
"""How long the two stages of a step take at B frames -- motion (keypoints, normalize_kp, dense motion) and net_g (+ uint8) --
alone, back to back on one stream, and with the motion stage of batch i+1 on a second stream beside net_g of batch i.
usage: python tools/stage_time.py [B] [f32|bf16]"""
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402
from synergize_motion_appearance_amd import driver, ops  # noqa: E402
from synergize_motion_appearance_amd.driver import normalize_kp  # noqa: E402
from synergize_motion_appearance_amd.synth import synth_clip  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 60
dev = torch.device("cuda", 0)
net_g, me, _, _ = bench.build_nets(dev)
if len(sys.argv) > 2 and sys.argv[2] == "bf16":
    net_g.set_compute_dtype("bf16"); me.set_compute_dtype("bf16")
src, drv = synth_clip(300, seed=123)
drv = drv.to(dev)
state = driver.encode_source_state(net_g, me, src.unsqueeze(0).to(dev), drv[0:1], True)
eng_g, eng_m = net_g.engine(), me.engine()


def motion(fr):
    kp_d = eng_m.estimate_kp(fr.float())
    kp_n = normalize_kp(state.kp_source, kp_d, state.kp_initial, True, True, True, state.scale)
    return eng_m.dense_motion(state.src64, kp_n, state.kp_source)


def netg(dm):
    st = eng_g.forward(state.cache, dm["deformation"], dm["occlusion_nhwc"].view(-1, 64, 64), dm["heat_nhwc"], 1.0)
    return ops.to_uint8(st["out"], -1.0, 1.0)


def wall(fn, n=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


batches = [drv[i * B:(i + 1) * B] for i in range(300 // B)]
dm0 = motion(batches[0])
t_m = wall(lambda: motion(batches[1]))
t_g = wall(lambda: netg(dm0))
t_seq = wall(lambda: [netg(motion(b)) for b in batches], n=2) / len(batches)
s_m = torch.cuda.Stream()
cur = torch.cuda.current_stream()


def piped():
    outs = []
    s_m.wait_stream(cur)
    for b in batches:
        with torch.cuda.stream(s_m):
            dm = motion(b)
            ev = torch.cuda.Event(); ev.record(s_m)
        cur.wait_event(ev)
        for v in dm.values():
            if torch.is_tensor(v):
                v.record_stream(cur)
        outs.append(netg(dm))
    return outs


t_pipe = wall(piped, n=2) / len(batches)
ref = [netg(motion(b)) for b in batches]
got = piped()
torch.cuda.synchronize()
same = all(torch.equal(a, b) for a, b in zip(ref, got))
print(f"B={B}: motion stage {t_m:.2f} ms, net_g stage {t_g:.2f} ms, sequential step {t_seq:.2f} ms ({1e3 * B / t_seq:.1f} fps), "
      f"two-stream step {t_pipe:.2f} ms ({1e3 * B / t_pipe:.1f} fps), identical frames: {same}")
