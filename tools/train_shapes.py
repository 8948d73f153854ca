#!/usr/bin/env python3
"""Per-shape table of the contraction launches of ONE eager training step (forward + backward, train.yml, B pairs): HIP events around
every convolution / Linear / weight-gradient launch (ops.profile()), grouped by (family, M, N, K, kernel size).
usage: python tools/train_shapes.py [--dtype f32|bf16] [--batch 4] [--top 40]"""
import argparse
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402
import yaml  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16"])
    ap.add_argument("--top", type=int, default=40)
    a = ap.parse_args()
    from basicsr.archs import build_network
    from synergize_motion_appearance_amd import ops
    from synergize_motion_appearance_amd.synth import synth_state_dict, synth_clip
    from synergize_motion_appearance_amd.trainer import TrainStep
    cfg = yaml.safe_load(open(os.path.join(REPO, "options/train.yml")))
    net_g, me = build_network(cfg["network_g"]), build_network(cfg["network_motion_estimator"])
    net_g.load_state_dict(synth_state_dict([(k, v.shape) for k, v in net_g.state_dict().items()]), strict=True)
    me.load_state_dict(synth_state_dict([(k, v.shape) for k, v in me.state_dict().items()]), strict=True)
    net_g, me = net_g.cuda(), me.cuda()
    topt = {k: v for k, v in cfg["train"].items() if k not in ("gan_opt",)}
    topt["perceptual_opt"] = dict(cfg["train"]["perceptual_opt"], synthetic_vgg19=True)
    topt["compute_dtype"] = a.dtype
    step = TrainStep(net_g, me, topt, use_graph=False, net_d=None)
    _, clip = synth_clip(2 * a.batch, seed=321)
    src, drv = clip[:a.batch].contiguous().cuda(), clip[a.batch:].contiguous().cuda()
    for _ in range(2):
        step.step(src, drv, gan=False)
    step.g.flat.zero_grad()
    step.flat_m.zero_grad()
    with ops.profile() as rec:
        step.forward_backward(src, drv, gan=False)
    agg = {}
    for name, meta, ms in rec.rows:
        meta = meta or {}
        key = (("winograd" if meta.get("wino") else name), meta.get("M"), meta.get("N"), meta.get("K"), meta.get("k"), meta.get("nb", 1))
        r = agg.setdefault(key, [0, 0.0, 0.0])
        r[0] += 1
        r[1] += ms
        r[2] += meta.get("flops", 0.0)
    tot = sum(r[1] for r in agg.values())
    print(f"# {a.dtype}, B={a.batch}: {len(rec.rows)} timed launches, {tot:.2f} ms; family M N K ksize nb : calls ms TF(algorithmic)")
    fam = {}
    for (n, *_), r in agg.items():
        f = fam.setdefault(n, [0, 0.0])
        f[0] += r[0]
        f[1] += r[1]
    for n, f in sorted(fam.items(), key=lambda t: -t[1][1]):
        print(f"#   {n:14s} {f[0]:5d} calls {f[1]:8.2f} ms")
    for key, r in sorted(agg.items(), key=lambda t: -t[1][1])[:a.top]:
        n, M, N, K, k, nb = key
        print(f"{n:12s} {M!s:>9} {N!s:>5} {K!s:>6} {k!s:>2} {nb!s:>3} : {r[0]:4d} {r[1]:8.3f} {r[2] / max(r[1], 1e-9) / 1e9:8.1f}")


if __name__ == "__main__":
    main()
