#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by importing the *reference* on CPU.

Runs ONLY in the build container (needs /root/reference); never on the GPU box and never
from a test.  The fixtures hold plain arrays only (inputs where they are not re-derivable
from a seed, and the reference's outputs) plus the state_dict name/shape manifest.
Weights are re-synthesised from names on both sides
(`synergize_motion_appearance_amd.synth`), inputs from seeds.

Import recipe: a namespace stub for `basicsr` so the package
__init__ (which star-imports data/losses/metrics) never runs; MagicMock for cv2 / imageio /
torchvision / ffmpeg (not touched on the hot path); torch.__version__ patched during
import because `basicsr/utils/misc.py:12-13` cannot parse '2.10.0+rocm7.0'.

usage: python tests/golden/make_golden.py
"""
import json
import os
import sys
import types
from unittest.mock import MagicMock

import numpy as np
import torch
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, REPO)

from synergize_motion_appearance_amd.synth import (synth_state_dict, synth_clip,  # noqa: E402
                                                   synth_keypoints, synth_input)


def load_reference():
    stub = types.ModuleType("basicsr")
    stub.__path__ = [os.path.join(REF, "basicsr")]
    sys.modules["basicsr"] = stub
    for m in ["cv2", "imageio", "ffmpeg", "torchvision", "torchvision.utils", "torchvision.models",
              "torchvision.models.vgg", "torchvision.transforms", "torchvision.transforms.functional",
              "lmdb"]:
        try:
            __import__(m)
        except Exception:
            sys.modules[m] = MagicMock()
    v = torch.__version__
    torch.__version__ = "2.10.0"
    try:
        from basicsr.archs import build_network
        import basicsr.demo as demo
    finally:
        torch.__version__ = v
    return build_network, demo


def np32(t):
    return t.detach().cpu().numpy()


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    build_network, demo = load_reference()
    from basicsr.utils.options import ordered_yaml
    from basicsr.utils import tensor2img
    cfg = yaml.load(open(os.path.join(REF, "options/test.yml")), Loader=ordered_yaml()[0])
    net_g = build_network(cfg["network_g"]).eval()
    me = build_network(cfg["network_motion_estimator"]).eval()

    # ---- manifest + name-keyed weights -------------------------------------------------
    man = {"network_g": [[k, list(v.shape)] for k, v in net_g.state_dict().items()],
           "network_motion_estimator": [[k, list(v.shape)] for k, v in me.state_dict().items()]}
    aa_ref = me.state_dict()["kp_detector.down.weight"].clone()
    net_g.load_state_dict(synth_state_dict(net_g.state_dict()), strict=True)
    me.load_state_dict(synth_state_dict(me.state_dict()), strict=True)
    assert torch.equal(aa_ref, me.state_dict()["kp_detector.down.weight"]), "antialias kernel restatement"
    json.dump(man, open(os.path.join(HERE, "manifest.json"), "w"))

    src, drv = synth_clip(8, seed=123)
    with torch.no_grad():
        # ---- A1-A3 keypoints ------------------------------------------------------------
        kp_s = me.estimate_kp(src[None])
        kp_d = me.estimate_kp(drv)
        np.savez_compressed(os.path.join(HERE, "kp.npz"),
                            src_value=np32(kp_s["value"]), src_jacobian=np32(kp_s["jacobian"]),
                            drv_value=np32(kp_d["value"]), drv_jacobian=np32(kp_d["jacobian"]))

        # ---- A0 normalize_kp: 4 flag combinations -----------------------------------------
        kp0 = {k: v[0:1] for k, v in kp_d.items()}
        kp3 = {k: v[3:4] for k, v in kp_d.items()}
        out = {}
        for rel in (False, True):
            for ad in (False, True):
                r = demo.normalize_kp(kp_source=kp_s, kp_driving={k: v.clone() for k, v in kp3.items()},
                                      kp_driving_initial=kp0, use_relative_movement=rel,
                                      use_relative_jacobian=rel, adapt_movement_scale=ad)
                out[f"value_r{int(rel)}a{int(ad)}"] = np32(r["value"])
                out[f"jacobian_r{int(rel)}a{int(ad)}"] = np32(r["jacobian"])
        np.savez_compressed(os.path.join(HERE, "normalize_kp.npz"), **out)

        # ---- A4-A6b dense motion (B=2: frames 2 and 5, absolute keypoints) ----------------
        idx = [2, 5]
        kpd2 = {k: v[idx] for k, v in kp_d.items()}
        kps2 = {k: v.repeat(2, 1, 1) if v.dim() == 3 else v.repeat(2, 1, 1, 1) for k, v in kp_s.items()}
        dm = me.estimate_motion_w_kp(kp_source=kps2, kp_driving=kpd2, source_image=src[None].repeat(2, 1, 1, 1))
        np.savez_compressed(os.path.join(HERE, "dense_motion.npz"),
                            deformation=np32(dm["deformation"]), occlusion_map=np32(dm["occlusion_map"]),
                            driving_kp_heatmap=np32(dm["driving_kp_heatmap"]),
                            mask=np32(dm["mask"][:, :, ::4, ::4]),
                            sparse_deformed=np32(dm["sparse_deformed"][:, :, :, ::4, ::4]))

        # ---- A7-A13 net_g on frame 2 (B=1) -------------------------------------------------
        dm1 = {k: (v[0:1] if torch.is_tensor(v) else v) for k, v in dm.items() if k not in ("kp_driving", "kp_source")}
        o = net_g(src[None], dm1, w=1, inference=True)
        sub = lambda t: np32(t[:, ::8, ::4, ::4])
        d = {"out": np32(o["out"]), "lq_feat": np32(o["lq_feat"])}
        for i, t in enumerate(o["out_occ"]):
            d[f"out_occ_{i}"] = np32(t)
        for i, t in enumerate(o["deformation_list"]):
            d[f"deformation_{i}"] = np32(t)
        for i, t in enumerate(o["res_deform_list"]):
            d[f"res_deform_{i}"] = np32(t)
        for key in ("deform_feat_list", "app_comp_list", "app_before_comp_list"):
            for i, t in enumerate(o[key]):
                d[f"{key}_{i}"] = sub(t)
        np.savez_compressed(os.path.join(HERE, "netg.npz"), **d)

        # ---- animate.py / model.test() side entries: encode_driving + generator on lq_feat ----------
        #      (models/appmotioncomp_model.py:450-454)
        ed = net_g.encode_driving(drv[2:3])
        aux = {f"enc_{k}": np32(v[:, ::8, ::4, ::4]) for k, v in ed.items()}
        aux["lq_recon"] = np32(net_g.generator(o["lq_feat"]))
        np.savez_compressed(os.path.join(HERE, "aux_entries.npz"), **aux)

        # ---- synthetic-keypoint mode: out-of-frame flow, zero padding, motion_ignore ------
        kps, kpd = synth_keypoints(2, seed=7)
        dms = me.estimate_motion_w_kp(kp_source=kps, kp_driving=kpd, source_image=src[None].repeat(2, 1, 1, 1))
        fl = dms["deformation"]
        frac_out = float(((fl > 1) | (fl < -1)).any(-1).float().mean())
        dms1 = {k: (v[1:2] if torch.is_tensor(v) else v) for k, v in dms.items() if k not in ("kp_driving", "kp_source")}
        os_ = net_g(src[None], dms1, w=1, inference=True)
        m32 = torch.nn.functional.interpolate(os_["deformation_list"][1].permute(0, 3, 1, 2), size=(32, 32),
                                              mode="bilinear", align_corners=True)
        ign = ((m32 > 1) | (m32 < -1)).any(1).reshape(-1)
        print(f"synthetic-kp: flow samples outside [-1,1]: {frac_out:.4f}; motion_ignore tokens @32: {int(ign.sum())}/1024")
        np.savez_compressed(os.path.join(HERE, "synthkp.npz"),
                            deformation=np32(dms["deformation"]), occlusion_map=np32(dms["occlusion_map"]),
                            driving_kp_heatmap=np32(dms["driving_kp_heatmap"][1:2]),
                            out=np32(os_["out"]), lq_feat=np32(os_["lq_feat"]),
                            out_occ_3=np32(os_["out_occ"][3]), deformation_4=np32(os_["deformation_list"][4]))

        # ---- config 1: demo.make_animation, 1 source + 8 frames ----------------------------
        e2e = {}
        preds, _ = demo.make_animation(src, list(drv), net_g, me, relative=False, adapt_movement_scale=False, cpu=True)
        e2e["frames_r0a0"] = np.stack(preds).astype(np.uint8)
        preds, _ = demo.make_animation(src, list(drv[:4]), net_g, me, relative=True, adapt_movement_scale=True, cpu=True)
        e2e["frames_r1a1"] = np.stack(preds).astype(np.uint8)
        kp7 = {k: v[7:8] for k, v in kp_d.items()}
        dm7 = me.estimate_motion_w_kp(kp_source=kp_s, kp_driving=kp7, source_image=src[None])
        e2e["out_f7"] = np32(net_g(src[None], dm7, w=1, inference=True)["out"])
        np.savez_compressed(os.path.join(HERE, "e2e.npz"), **e2e)

        # ---- A12 VectorQuantizer (train-only in the reference; micro-benchmark kernel) ------
        vq = {}
        for tag, quant, D, scale in (("m256", net_g.quantize_motion, 32, 0.25), ("m1024", net_g.quantize_motion, 32, 1.0),
                                     ("a512", net_g.quantize_app, 256, 0.5), ("a1024", net_g.quantize_app, 256, None)):
            z = synth_input(f"vq_{tag}", (2, D, 32, 32))
            zq, loss, st = quant(z, scale) if scale is not None else quant(z)
            vq[f"{tag}_indices"] = st["min_encoding_indices"].numpy().astype(np.int64)
            vq[f"{tag}_loss"] = np32(loss)
            vq[f"{tag}_zq_sub"] = np32(zq[:, :, ::4, ::4])
            vq[f"{tag}_perplexity"] = np32(st["perplexity"])
            vq[f"{tag}_mean_distance"] = np32(st["mean_distance"])
        np.savez_compressed(os.path.join(HERE, "vq.npz"), **vq)

        # ---- A14 tensor2img ---------------------------------------------------------------
        t = synth_input("tensor2img", (3, 64, 64)) * 0.8
        np.savez_compressed(os.path.join(HERE, "tensor2img.npz"),
                            img=tensor2img([t[None]], rgb2bgr=False, min_max=(-1, 1)))
    for f in sorted(os.listdir(HERE)):
        print(f, os.path.getsize(os.path.join(HERE, f)))


if __name__ == "__main__":
    main()
