#!/usr/bin/env python3
"""Time the generator + motion-estimator training step (BASELINE configs[4]: train.yml, B pairs per GPU, fp32) on one MI355X.
usage: python tools/train_bench.py [--batch 4] [--steps 5] [--warmup 2]"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402
import yaml  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--perceptual", action="store_true", help="+ MultiScalePyramidPerceptualLoss on out and out_lr (VGG19 layout, synthetic weights)")
    ap.add_argument("--gan", action="store_true", help="the form past net_d_start_iter: discriminator term with the adaptive weight + the discriminator's own step")
    ap.add_argument("--graph", action="store_true", help="train.use_hip_graph: replay the captured step")
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16"], help="bf16: convolution / Linear contractions on the bf16 MFMA (train.compute_dtype)")
    ap.add_argument("--tune", action="append", default=[], help="name=value for smx_set_tuning before anything is launched or captured (A/B runs)")
    a = ap.parse_args()
    if a.tune:
        from synergize_motion_appearance_amd import ops as _ops
        for kv in a.tune:
            k, v = kv.split("=")
            _ops.set_tuning(k, int(v))
    from basicsr.archs import build_network
    from synergize_motion_appearance_amd.synth import synth_state_dict, synth_clip
    from synergize_motion_appearance_amd.trainer import TrainStep
    cfg = yaml.safe_load(open(os.path.join(REPO, "options/train.yml")))
    net_g, me = build_network(cfg["network_g"]), build_network(cfg["network_motion_estimator"])
    net_g.load_state_dict(synth_state_dict([(k, v.shape) for k, v in net_g.state_dict().items()]), strict=True)
    me.load_state_dict(synth_state_dict([(k, v.shape) for k, v in me.state_dict().items()]), strict=True)
    net_g, me = net_g.cuda(), me.cuda()
    topt = {k: v for k, v in cfg["train"].items() if k not in ("perceptual_opt",) + (() if a.gan else ("gan_opt",))}
    net_d = None
    if a.gan:
        net_d = build_network(cfg["network_d"])
        net_d.load_state_dict(synth_state_dict([(k, v.shape) for k, v in net_d.state_dict().items()]), strict=True)
        net_d = net_d.cuda()
    if a.perceptual:
        topt["perceptual_opt"] = dict(cfg["train"]["perceptual_opt"], synthetic_vgg19=True)
    topt["compute_dtype"] = a.dtype
    step = TrainStep(net_g, me, topt, use_graph=a.graph, net_d=net_d)
    _, clip = synth_clip(2 * a.batch, seed=321)
    src, drv = clip[:a.batch].contiguous().cuda(), clip[a.batch:].contiguous().cuda()
    for _ in range(a.warmup + (3 if a.graph else 0)):
        step.step(src, drv, gan=a.gan)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        losses, _ = step.step(src, drv, gan=a.gan)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    print(json.dumps({"what": f"train.yml generator + motion-estimator step ({'with' if a.perceptual else 'no'} perceptual loss, {'with the GAN branch' if a.gan else 'no GAN'}), compute {a.dtype}{', hipGraph replay' if a.graph else ''}, 1 GPU", "batch": a.batch, "ms_per_step": round(1e3 * dt, 2),
                      "pairs_per_s": round(a.batch / dt, 2), "l_g_total": float(losses["l_g_total"]),
                      "peak_mem_GB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)}))


if __name__ == "__main__":
    main()
