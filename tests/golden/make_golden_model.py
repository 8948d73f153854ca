#!/usr/bin/env python3
"""Golden fixture for SURVEY.md section 8(f) row N1: the reference's MODEL-level animation entry,
`AppMotionCompModel.generate_video_image` / `.make_animation`
(/root/reference/basicsr/models/appmotioncomp_model.py:607-756), run here on CPU.

Runs ONLY in the build container.  What is executed is the reference's own class; around it:
  * `basicsr`, `basicsr.models` namespace stubs so the package __init__s (which import every
    model / loss / metric / dataset file) never run; MagicMock for imageio / ffmpeg / torchvision /
    lmdb / flow_vis / basicsr.losses / basicsr.metrics (none is touched by these two methods);
  * cv2 is ABSENT from this image.  The two methods reach it only through
    `tensor2img(..., rgb2bgr=True)` -> `cv2.cvtColor(img, COLOR_RGB2BGR)` and the final
    `cv2.cvtColor(p, COLOR_BGR2RGB)`: the cv2 mock's `cvtColor` is given the channel reversal
    those two codes mean (`img[..., ::-1]`).  DISCLOSED STAND-IN: nothing else of cv2 is emulated;
  * the module's `imwrite` / `mimsave` are replaced by recorders (they would call cv2 / imageio),
    so the fixture holds exactly the arrays and relative paths the reference would have written;
  * `build_network` is wrapped so the motion estimator the method constructs gets the same
    name-keyed synthetic weights as everywhere else (the yml's checkpoint paths do not exist).

usage: cd /tmp && python /root/repo/tests/golden/make_golden_model.py
"""
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

from synergize_motion_appearance_amd.synth import synth_state_dict, synth_clip  # noqa: E402

N_FRAMES, ANCHOR, SEED = 5, 2, 321


def load_reference_model():
    for name in ("basicsr", "basicsr.models"):
        stub = types.ModuleType(name)
        stub.__path__ = [os.path.join(REF, *name.split("."))]
        sys.modules[name] = stub
    for m in ["cv2", "imageio", "ffmpeg", "torchvision", "torchvision.utils", "torchvision.models",
              "torchvision.models.vgg", "torchvision.transforms", "torchvision.transforms.functional",
              "lmdb", "flow_vis", "basicsr.losses", "basicsr.metrics"]:
        try:
            __import__(m)
        except Exception:
            sys.modules[m] = MagicMock()
    cv2 = sys.modules["cv2"]
    assert isinstance(cv2, MagicMock), "a real cv2 appeared: drop the stand-in"
    cv2.cvtColor = lambda img, code: np.ascontiguousarray(img[..., ::-1])   # RGB2BGR / BGR2RGB only
    v = torch.__version__
    torch.__version__ = "2.10.0"
    try:
        import basicsr.models.appmotioncomp_model as M
    finally:
        torch.__version__ = v
    return M


class FakeLoader:
    """what `generate_video_image` reads from its dataloader (batch size 1 collation of
    data/frames_dataset.py:244-306: tensors gain a leading 1, strings become 1-lists)."""

    def __init__(self, items):
        self.items = items
        self.dataset = types.SimpleNamespace(opt={"name": "synthetic"})

    def __len__(self):
        return len(self.items)

    def __iter__(self):
        return iter(self.items)


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    M = load_reference_model()
    from basicsr.utils.options import ordered_yaml
    cfg = yaml.load(open(os.path.join(REF, "options/test.yml")), Loader=ordered_yaml()[0])
    opt = dict(cfg)
    opt.update(num_gpu=0, is_train=False, dist=False, rank=0)
    opt["path"] = {"pretrain_network_g": None, "pretrain_network_motion_estimator": None, "visualization": "VIS"}
    opt["val"] = {"relative": True, "adapt_scale": True, "w": 1, "metrics": None}

    real_build = M.build_network

    def build_with_synth(o):
        net = real_build(o)
        net.load_state_dict(synth_state_dict(net.state_dict()), strict=True)
        return net
    M.build_network = build_with_synth
    import basicsr.models.sr_model as SR
    SR.build_network = build_with_synth
    written, videos = [], []
    M.imwrite = lambda img, path, *a, **k: written.append((path, np.array(img)))
    M.mimsave = lambda frames, path, *a, **k: videos.append((path, np.stack([np.array(f) for f in frames])))

    model = M.AppMotionCompModel(opt)
    src, drv = synth_clip(N_FRAMES, seed=SEED)
    item = {"source": src[None], "driving_video": [f[None] for f in drv], "anchor_idx": ANCHOR,
            "video_name": ["clip0"], "driving_name_list": [[f"{i:04d}"] for i in range(N_FRAMES)]}
    model.generate_video_image(FakeLoader([item]), current_iter="golden", tb_logger=None)

    res = [(p, a) for p, a in written if p.endswith("_r.png")]
    vis = [(p, a) for p, a in written if p.endswith("_v.png")]
    assert len(res) == N_FRAMES and len(vis) == N_FRAMES and len(videos) == 2
    result_png, visual_png = np.stack([a for _, a in res]), np.stack([a for _, a in vis])
    # the model's own make_animation on the forward half (BGR uint8 lists)
    preds, drvs = model.make_animation(src[None], [f[None] for f in drv[ANCHOR:]])
    # redundancy checked here so the fixture can stay small: the mp4 frames are the RGB view of the
    # PNG arrays, the visual strip is [source | driving | prediction], make_animation(forward half)
    # is the tail of the spliced list
    assert videos[0][0].endswith("_r.mp4") and np.array_equal(videos[0][1][..., ::-1], result_png)
    assert videos[1][0].endswith("_v.mp4") and np.array_equal(videos[1][1][..., ::-1], visual_png)
    assert np.array_equal(visual_png[:, :, 512:], result_png)
    assert np.array_equal(np.stack(preds), result_png[ANCHOR:])
    assert np.array_equal(np.stack(drvs), visual_png[ANCHOR:, :, 256:512])
    out = {"result_png": result_png, "visual_png_0": visual_png[0],
           "all_paths": np.array([p for p, _ in written]), "video_paths": np.array([p for p, _ in videos]),
           "n_frames": N_FRAMES, "anchor_idx": ANCHOR, "seed": SEED}
    np.savez_compressed(os.path.join(HERE, "model_animate.npz"), **out)
    print("wrote model_animate.npz:", {k: getattr(v, "shape", v) for k, v in out.items()})


if __name__ == "__main__":
    main()
