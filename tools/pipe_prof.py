#!/usr/bin/env python3
"""where does FramePipeline's per-batch host time go?  eager render_frames alone vs inside the pipeline, copy streams separate vs shared."""
import os, sys, time
import torch, yaml
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from basicsr.archs import build_network
from synergize_motion_appearance_amd import driver, ops
from synergize_motion_appearance_amd.synth import synth_state_dict, synth_clip
cfg = yaml.safe_load(open(os.path.join(REPO, "options/test.yml")))
net_g, me = build_network(cfg["network_g"]), build_network(cfg["network_motion_estimator"])
net_g.load_state_dict(synth_state_dict([(k, v.shape) for k, v in net_g.state_dict().items()]))
me.load_state_dict(synth_state_dict([(k, v.shape) for k, v in me.state_dict().items()]))
net_g, me = net_g.cuda().eval(), me.cuda().eval()
src, drv = synth_clip(64, seed=1)
u8 = ops.to_uint8(drv.cuda().permute(0, 2, 3, 1).contiguous(), -1.0, 1.0).cpu().pin_memory()
st = driver.encode_source_state(net_g, me, src.cuda(), drv[0:1].cuda(), True)
B = 4
x = ops.frames_u8_to_nchw(u8.cuda())
def timed(fn, n=2):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t0) / n
print("render_frames alone, 16 batches of 4:", round(timed(lambda: driver.render_frames(st, x, net_g, me, True, True, batch=B)), 1), "ms")
with torch.no_grad():
    print("  ... under no_grad:", round(timed(lambda: driver.render_frames(st, x, net_g, me, True, True, batch=B)), 1), "ms")
side = torch.cuda.Stream()
def with_side():
    for i in range(0, 64, B):
        with torch.cuda.stream(side):
            y = u8[i:i + B].cuda(non_blocking=True); ev = torch.cuda.Event(); ev.record()
        torch.cuda.current_stream().wait_event(ev)
        driver.render_frames(st, x[i:i + B], net_g, me, True, True, batch=B)
print("  ... with an H2D on a side stream + wait_event per batch:", round(timed(with_side), 1), "ms")
out = torch.empty((64, 256, 256, 3), dtype=torch.uint8).pin_memory()
for ug in (False, True):
    pipe = driver.FramePipeline(net_g, me, batch=B, use_graph=ug)
    print(f"pipeline use_graph={ug}:", round(timed(lambda: pipe.run(st, u8, out)), 1), "ms")
    cur = torch.cuda.current_stream()
    pipe.s_h2d = pipe.s_d2h = cur
    print(f"  ... copies on the compute stream:", round(timed(lambda: pipe.run(st, u8, out)), 1), "ms")

# phase timing of the pipeline loop (host times, no extra synchronisation)
import collections
pipe = driver.FramePipeline(net_g, me, batch=B, use_graph=False)
acc = collections.defaultdict(float)
orig_render = pipe._render
def timed_render(state, u8d):
    t = time.perf_counter(); r = orig_render(state, u8d); acc["render(host)"] += time.perf_counter() - t; return r
pipe._render = timed_render
orig_sync = torch.cuda.Event.synchronize
def ev_sync(self):
    t = time.perf_counter(); orig_sync(self); acc["event.synchronize"] += time.perf_counter() - t
torch.cuda.Event.synchronize = ev_sync
pipe.run(st, u8, out); torch.cuda.synchronize()
acc.clear(); t0 = time.perf_counter(); pipe.run(st, u8, out); torch.cuda.synchronize()
print("phases over 16 batches:", {k: round(1e3 * v, 1) for k, v in acc.items()}, "total", round(1e3 * (time.perf_counter() - t0), 1))

torch.cuda.Event.synchronize = orig_sync
def loop_sync_each():
    for i in range(0, 64, B):
        driver.render_frames(st, x[i:i + B], net_g, me, True, True, batch=B)
        torch.cuda.synchronize()
print("render_frames, device sync after every batch:", round(timed(loop_sync_each), 1), "ms")
def loop_sync_lag1():
    evs = []
    for i in range(0, 64, B):
        driver.render_frames(st, x[i:i + B], net_g, me, True, True, batch=B)
        e = torch.cuda.Event(); e.record(); evs.append(e)
        if len(evs) > 1: evs[-2].synchronize()
print("render_frames, wait for the PREVIOUS batch after launching each:", round(timed(loop_sync_lag1), 1), "ms")
def loop_host_copy():
    for i in range(0, 64, B):
        y = ops.frames_u8_to_nchw(u8[i:i + B].cuda(non_blocking=True))
        driver.render_frames(st, y, net_g, me, True, True, batch=B)
print("render_frames + H2D + normalise per batch (same stream):", round(timed(loop_host_copy), 1), "ms")

pin = torch.empty((B, 256, 256, 3), dtype=torch.uint8).pin_memory()
def loop_d2h_side():
    for i in range(0, 64, B):
        o = driver.render_frames(st, x[i:i + B], net_g, me, True, True, batch=B)
        ev = torch.cuda.Event(); ev.record()
        with torch.cuda.stream(side):
            side.wait_event(ev); pin.copy_(o, non_blocking=True); o.record_stream(side)
print("render_frames + D2H of the result on a side stream:", round(timed(loop_d2h_side), 1), "ms")
def loop_d2h_same():
    for i in range(0, 64, B):
        o = driver.render_frames(st, x[i:i + B], net_g, me, True, True, batch=B)
        pin.copy_(o, non_blocking=True)
print("render_frames + D2H on the compute stream:", round(timed(loop_d2h_same), 1), "ms")
pg = torch.empty((B, 256, 256, 3), dtype=torch.uint8)
def loop_d2h_pageable():
    for i in range(0, 64, B):
        o = driver.render_frames(st, x[i:i + B], net_g, me, True, True, batch=B)
        pg.copy_(o)
print("render_frames + blocking D2H to pageable memory:", round(timed(loop_d2h_pageable), 1), "ms")

pipe = driver.FramePipeline(net_g, me, batch=B, use_graph=False)
du8 = u8.cuda()
def loop_render_direct():
    for i in range(0, 64, B):
        pipe._render(st, du8[i:i + B])
print("pipe._render directly, 16 batches:", round(timed(loop_render_direct), 1), "ms")
def loop_stream_only():
    for a, chunk in pipe.stream(st, u8):
        pass
print("pipe.stream without the host copy-out:", round(timed(loop_stream_only), 1), "ms")
import torch.utils._contextlib
def gen():
    for i in range(0, 64, B):
        driver.render_frames(st, x[i:i + B], net_g, me, True, True, batch=B)
        yield i
def loop_gen_plain():
    for _ in gen(): pass
print("render_frames inside a plain generator:", round(timed(loop_gen_plain), 1), "ms")
ng = torch.no_grad()(gen)
def loop_gen_nograd():
    for _ in ng(): pass
print("render_frames inside a @torch.no_grad() generator:", round(timed(loop_gen_nograd), 1), "ms")

# bisect: rebuild the pipeline loop piece by piece
s_h2d, s_d2h = torch.cuda.Stream(), torch.cuda.Stream()
pin_in = [torch.empty((B, 256, 256, 3), dtype=torch.uint8).pin_memory() for _ in range(2)]
dev_in = [torch.empty((B, 256, 256, 3), dtype=torch.uint8, device="cuda") for _ in range(2)]
pin_out = [torch.empty((B, 256, 256, 3), dtype=torch.uint8).pin_memory() for _ in range(2)]
def variant(pinned_copy, h2d_side, d2h_side, sync_prev, in_free_sync):
    cur = torch.cuda.current_stream()
    in_free, out_ready, pending = [None, None], [None, None], None
    for bi, i in enumerate(range(0, 64, B)):
        k = bi & 1
        if in_free_sync and in_free[k] is not None: in_free[k].synchronize()
        if pinned_copy: pin_in[k].copy_(u8[i:i + B])
        if h2d_side:
            with torch.cuda.stream(s_h2d):
                dev_in[k].copy_(pin_in[k], non_blocking=True); h = torch.cuda.Event(); h.record()
            cur.wait_event(h)
        o = pipe._render(st, dev_in[k] if h2d_side else du8[i:i + B])
        in_free[k] = torch.cuda.Event(); in_free[k].record(cur)
        if d2h_side:
            done = torch.cuda.Event(); done.record(cur)
            with torch.cuda.stream(s_d2h):
                s_d2h.wait_event(done); pin_out[k].copy_(o, non_blocking=True); o.record_stream(s_d2h)
                out_ready[k] = torch.cuda.Event(); out_ready[k].record()
            if sync_prev and pending is not None: out_ready[pending].synchronize()
            pending = k
for name, args in (("pinned host copy only", (1, 0, 0, 0, 0)), ("+ H2D on side stream", (1, 1, 0, 0, 0)), ("+ D2H on side stream", (1, 1, 1, 0, 0)),
                   ("+ wait for previous batch's D2H", (1, 1, 1, 1, 0)), ("+ in_free sync", (1, 1, 1, 1, 1))):
    print(f"  {name}:", round(timed(lambda: variant(*args)), 1), "ms")
