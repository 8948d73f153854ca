"""the MODEL / dataset / animate.py entry around the hot path.

CPU tests: host logic (PNG codec, cv2-free image I/O conventions, dataset dict, parse(), registry,
array metrics).  GPU tests: `AppMotionCompModel.generate_video_image` against the fixture produced
by running the REFERENCE's own class (tests/golden/make_golden_model.py) and the dataset-driven
`basicsr/animate.py` pipeline end to end."""
import os
import types

import numpy as np
import pytest
import torch
import yaml

from tests.util import golden, weights, clip, HERE
from synergize_motion_appearance_amd import img_util as U
from synergize_motion_appearance_amd.png import encode_png, decode_png

REPO = os.path.dirname(HERE)


# ---------------------------------------------------------------- CPU: host logic
def test_png_codec_roundtrip_and_all_filter_types():
    import struct
    import zlib
    from synergize_motion_appearance_amd.png import _chunk, _SIG
    rng = np.random.default_rng(0)
    for shp in [(17, 23, 3), (8, 8), (5, 9, 4), (4, 4, 1), (1, 1, 3)]:
        a = rng.integers(0, 256, shp, dtype=np.uint8)
        b = decode_png(encode_png(a))
        assert np.array_equal(b, a[:, :, 0] if a.ndim == 3 and a.shape[2] == 1 else a), shp

    def enc_filter(a, ft):                                  # PNG spec section 9 filters, written out
        h, w, c = a.shape
        s = w * c
        raw = np.zeros((h, 1 + s), np.uint8)
        prev = np.zeros(s, np.int32)
        for y in range(h):
            cur = a[y].reshape(-1).astype(np.int32)
            for x in range(s):
                A = cur[x - c] if x >= c else 0
                B = prev[x]
                C = prev[x - c] if x >= c else 0
                if ft == 4:
                    pp = A + B - C
                    pa, pb, pc = abs(pp - A), abs(pp - B), abs(pp - C)
                    p = A if pa <= pb and pa <= pc else (B if pb <= pc else C)
                else:
                    p = [0, A, B, (A + B) >> 1][ft]
                raw[y, 1 + x] = (cur[x] - p) & 255
            raw[y, 0] = ft
            prev = cur
        return (_SIG + _chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0))
                + _chunk(b"IDAT", zlib.compress(raw.tobytes())) + _chunk(b"IEND", b""))
    a = rng.integers(0, 256, (11, 13, 3), dtype=np.uint8)
    for ft in range(5):
        assert np.array_equal(decode_png(enc_filter(a, ft)), a), ft
    with pytest.raises(ValueError):
        decode_png(b"not a png")
    with pytest.raises(TypeError):
        encode_png(a.astype(np.float32))


def test_image_io_follows_the_cv2_bgr_convention(tmp_path):
    rng = np.random.default_rng(1)
    bgr = rng.integers(0, 256, (6, 7, 3), dtype=np.uint8)
    p = str(tmp_path / "sub" / "x.png")
    assert U.imwrite(bgr, p)                                # auto_mkdir
    raw = open(p, "rb").read()
    assert np.array_equal(decode_png(raw), bgr[:, :, ::-1])  # RGB in the file
    assert np.array_equal(U.imfrombytes(raw), bgr)           # BGR in memory
    f = U.imfrombytes(raw, float32=True)
    assert f.dtype == np.float32 and np.allclose(f, bgr / 255.0)
    with pytest.raises(ValueError):
        U.imwrite(bgr, str(tmp_path / "x.jpg"))
    # resize: identity at equal size, exact on a linear ramp when upsampling x2 (interior)
    ramp = np.tile(np.arange(8, dtype=np.float32)[None, :, None], (8, 1, 3))
    assert np.array_equal(U.resize_linear(ramp, (8, 8)), ramp)
    up = U.resize_linear(ramp, (16, 16))
    assert up.shape == (16, 16, 3) and np.allclose(up[0, 1:-1, 0], (np.arange(1, 15) + 0.5) / 2 - 0.5)
    d = U.mimsave([bgr, bgr], str(tmp_path / "v" / "a.mp4"))
    if isinstance(d, str):                                   # no imageio here: numbered PNG frames
        assert sorted(os.listdir(d)) == ["000000.png", "000001.png"]


def test_array_metrics():
    from synergize_motion_appearance_amd.models import calculate_psnr, calculate_l1, calculate_ssim
    rng = np.random.default_rng(2)
    a = rng.integers(0, 256, (32, 32, 3), dtype=np.uint8)
    b = a.copy()
    b[0, 0, 0] = (int(b[0, 0, 0]) + 16) % 256
    assert calculate_psnr(a, a) == float("inf") and calculate_l1(a, a) == 0.0
    assert abs(calculate_ssim(a, a) - 1.0) < 1e-12
    mse = np.mean((a.astype(np.float64) - b) ** 2)
    assert abs(calculate_psnr(a, b) - 10 * np.log10(255 ** 2 / mse)) < 1e-9
    assert abs(calculate_l1(a, b) - np.abs(a.astype(np.float64) - b).mean()) < 1e-12
    assert calculate_ssim(a, 255 - a) < 0.1
    assert calculate_psnr(a, b, crop_border=2) == float("inf")   # the changed pixel is cropped away


def _write_clip(root, n=4, seed=11, size=256):
    """a source frame + a driving folder of PNGs (uint8-quantised synthetic clip) + the pairs csv."""
    from synergize_motion_appearance_amd.synth import synth_clip
    src, drv = synth_clip(n, seed=seed)

    def to_bgr8(t):
        return np.ascontiguousarray(((t.clamp(-1, 1) + 1) * 127.5).round().byte().permute(1, 2, 0).numpy()[:, :, ::-1])
    sdir, ddir = os.path.join(root, "id01.mp4"), os.path.join(root, "id02.mp4")
    U.imwrite(to_bgr8(src), os.path.join(sdir, "0000.png"))
    for i in range(n):
        U.imwrite(to_bgr8(drv[i]), os.path.join(ddir, f"{i:04d}.png"))
    csv = os.path.join(root, "pairs.csv")
    with open(csv, "w") as f:
        f.write("source,driving,anchor,anchor_idx\n")
        f.write(f"{os.path.join(sdir, '0000.png')},{ddir},{os.path.join(ddir, '0002.png')},1\n")
    return csv, src, drv


def test_anchor_dataset_dict_and_reference_quirks(tmp_path):
    from basicsr.data import build_dataset, build_dataloader
    from basicsr.utils import DATASET_REGISTRY
    csv, src, drv = _write_clip(str(tmp_path), n=4)
    opt = {"name": "synthetic", "type": "FramesMotionTransferTestDataset_CrossID_videopair_anchor", "phase": "test",
           "root_dir": str(tmp_path), "pairs_list": csv, "gt_size": 256, "io_backend": {"type": "disk"}}
    ds = build_dataset(opt)
    assert len(ds) == 1 and "FramesMotionTransferTestDataset_CrossID_videopair_anchor" in DATASET_REGISTRY
    it = ds[0]
    assert set(it) == {"source", "driving_video", "anchor", "video_name", "driving_name_list", "anchor_idx"}
    assert it["video_name"] == "id01_0000_id02"             # [:-4] of folder, file and driving folder
    assert it["driving_name_list"] == ["0001.png", "0002.png", "0003.png"]   # frame 0 of the folder is skipped
    assert it["anchor_idx"] == 1 and it["source"].shape == (3, 256, 256) and len(it["driving_video"]) == 3
    q = lambda t: ((t.clamp(-1, 1) + 1) * 127.5).round() / 127.5 - 1   # noqa: E731
    assert float((it["source"] - q(src)).abs().max()) < 1e-6 and float((it["driving_video"][0] - q(drv[1])).abs().max()) < 1e-6
    assert torch.equal(it["anchor"], it["driving_video"][1])  # anchor frame 0002.png == kept frame index 1
    csv2 = str(tmp_path / "pairs_noanchor.csv")               # no anchor column: driving[0] and index 0 win,
    lines = open(csv).read().splitlines()                     # whatever anchor_idx says (frames_dataset.py:258-262)
    open(csv2, "w").write("source,driving,anchor_idx\n" + ",".join(lines[1].split(",")[:2]) + ",1\n")
    it2 = build_dataset(dict(opt, pairs_list=csv2))[0]
    assert it2["anchor_idx"] == 0 and torch.equal(it2["anchor"], it2["driving_video"][0])
    batch = next(iter(build_dataloader(ds, opt)))
    assert batch["source"].shape == (1, 3, 256, 256) and batch["video_name"] == ["id01_0000_id02"]
    assert batch["driving_name_list"][2][0] == "0003.png" and int(batch["anchor_idx"]) == 1
    ds2 = build_dataset(dict(opt, gt_size=128, max_frame=1))
    assert ds2[0]["source"].shape == (3, 128, 128) and len(ds2[0]["driving_video"]) == 1
    with pytest.raises(ValueError):
        build_dataloader(ds, dict(opt, phase="train"))
    with pytest.raises(NotImplementedError):
        build_dataset(dict(opt, pairs_list=None))


def test_parse_and_model_registry(tmp_path):
    from basicsr.utils.options import parse, dict2str
    from basicsr.models import build_model
    from basicsr.utils import MODEL_REGISTRY
    opt = parse(os.path.join(REPO, "options/test.yml"), str(tmp_path), is_train=False)
    assert opt["is_train"] is False and opt["name"].endswith("_appearance-motion-compensation")
    assert opt["path"]["visualization"] == os.path.join(opt["path"]["results_root"], "visualization")
    assert opt["path"]["results_root"].startswith(os.path.join(opt["path"]["save_path"], "results"))
    assert "network_g:[" in dict2str(opt) and opt["model_type"] in MODEL_REGISTRY
    # training ymls (row N2): experiments/<name>/{models, training_states, visualization}
    topt = parse(os.path.join(REPO, "options/train.yml"), str(tmp_path), is_train=True)
    assert topt["is_train"] is True and topt["path"]["models"] == os.path.join(topt["path"]["experiments_root"], "models")
    assert topt["path"]["experiments_root"] == os.path.join("./train_log", "experiments", topt["name"])
    assert topt["path"]["training_states"].endswith("training_states") and topt["train"]["optim_g"]["lr"] == 8e-5
    assert topt["datasets"]["train"]["phase"] == "train" and topt["train"]["net_d_start_iter"] == 5001
    with pytest.raises(KeyError):
        build_model({"model_type": "NoSuchModel"})
    with pytest.raises(RuntimeError):                        # no CPU mode, and it says so
        build_model(dict(opt, num_gpu=0))


def _write_pairs(root, n=3, seed=11):
    """frame-pair csv for FramesMotionTransferTestDataset_PairsList on top of the clip files of `_write_clip`."""
    _, src, drv = _write_clip(root, n=n, seed=seed)
    sdir, ddir = os.path.join(root, "id01.mp4"), os.path.join(root, "id02.mp4")
    csv = os.path.join(root, "frame_pairs.csv")
    with open(csv, "w") as f:
        f.write("source,driving\n")
        for i in range(1, n):
            f.write(f"{os.path.join(sdir, '0000.png')},{os.path.join(ddir, f'{i:04d}.png')}\n")
    return csv, src, drv


def test_pairs_list_dataset(tmp_path):
    """`test.py`'s frame-pair dataset (reference data/frames_dataset.py:309-399): dict keys, the 4-component frame_name, anchor falling
    back to the driving frame, per-frame resize to gt_size."""
    from basicsr.data import build_dataset, build_dataloader
    csv, src, drv = _write_pairs(str(tmp_path), n=3)
    opt = {"name": "pairs", "type": "FramesMotionTransferTestDataset_PairsList", "phase": "test", "root_dir": str(tmp_path),
           "pairs_list": csv, "gt_size": 256, "io_backend": {"type": "disk"}}
    ds = build_dataset(opt)
    assert len(ds) == 2
    it = ds[1]
    assert set(it) == {"source", "driving", "anchor", "frame_name"} and it["frame_name"] == "id01_0000_id02_0002"
    assert it["source"].shape == (3, 256, 256) and torch.equal(it["anchor"], it["driving"])
    q = lambda t: ((t.clamp(-1, 1) + 1) * 127.5).round() / 127.5 - 1                    # noqa: E731  (the PNGs hold uint8-quantised frames)
    assert float((it["driving"] - q(drv[2])).abs().max()) < 1e-6 and float((it["source"] - q(src)).abs().max()) < 1e-6
    small = build_dataset(dict(opt, gt_size=128))[0]
    assert small["source"].shape == (3, 128, 128) and small["driving"].shape == (3, 128, 128)
    batch = next(iter(build_dataloader(ds, opt)))
    assert batch["source"].shape == (1, 3, 256, 256) and batch["frame_name"] == ["id01_0000_id02_0001"]
    with pytest.raises(NotImplementedError):
        build_dataset({k: v for k, v in opt.items() if k != "pairs_list"})


# ---------------------------------------------------------------- GPU: parity with the reference's class
class FakeLoader:
    def __init__(self, items):
        self.items, self.dataset = items, types.SimpleNamespace(opt={"name": "synthetic"})

    def __len__(self):
        return len(self.items)

    def __iter__(self):
        return iter(self.items)


def _model(tmp_path, val):
    from basicsr.models import build_model
    cfg = yaml.safe_load(open(os.path.join(REPO, "options/test.yml")))
    cfg.update(is_train=False, dist=False, rank=0)
    cfg["path"] = {"pretrain_network_g": None, "pretrain_network_motion_estimator": None, "visualization": str(tmp_path / "VIS")}
    cfg["val"] = val
    model = build_model(cfg)
    model.net_g.load_state_dict(weights("network_g"), strict=True)
    model._ensure_motion_estimator().load_state_dict(weights("network_motion_estimator"), strict=True)
    return model


@pytest.mark.gpu
def test_generate_video_image_matches_the_reference_class(tmp_path):
    g = golden("model_animate.npz")
    n, anchor, seed = int(g["n_frames"]), int(g["anchor_idx"]), int(g["seed"])
    model = _model(tmp_path, {"relative": True, "adapt_scale": True, "w": 1, "metrics": None, "batch": 2})
    src, drv = clip(n, seed)
    item = {"source": src[None], "driving_video": [f[None] for f in drv], "anchor_idx": anchor,
            "video_name": ["clip0"], "driving_name_list": [[f"{i:04d}"] for i in range(n)]}
    model.generate_video_image(FakeLoader([item]), current_iter="golden", tb_logger=None)
    root = str(tmp_path)
    want_paths = [str(p) for p in g["all_paths"]]
    for p in want_paths:                                     # same files at the same relative paths
        assert os.path.exists(os.path.join(root, p)), p
    got = np.stack([U.imfrombytes(open(os.path.join(root, p), "rb").read()) for p in want_paths if p.endswith("_r.png")])
    ref = g["result_png"]
    diff = np.abs(got.astype(np.int16) - ref.astype(np.int16))
    assert diff.max() <= 1, diff.max()                       # uint8 frames within 1 LSB of the reference's
    assert (diff > 0).mean() < 2e-3, (diff > 0).mean()       # ... and almost everywhere identical
    v0 = U.imfrombytes(open(os.path.join(root, want_paths[0]), "rb").read())
    assert np.abs(v0.astype(np.int16) - g["visual_png_0"].astype(np.int16)).max() <= 1
    # [source | driving] panels: exactly the reference's tensor2img of the inputs (the synthetic clip
    # itself is only reproducible to an ulp across hosts, hence not compared bit-for-bit with the fixture)
    assert np.array_equal(v0[:, :256], U.tensor2img([src[None].clone()], rgb2bgr=True, min_max=(-1, 1)))
    assert np.array_equal(v0[:, 256:512], U.tensor2img([drv[0][None].clone()], rgb2bgr=True, min_max=(-1, 1)))
    for p in g["video_paths"]:
        p = os.path.join(root, str(p))
        assert os.path.exists(p) or os.path.isdir(p + ".frames"), p
    # the model's own make_animation on the forward half == tail of the spliced list (both BGR)
    preds, drvs = model.make_animation(src[None], [f[None] for f in drv[anchor:]])
    assert len(preds) == n - anchor and np.abs(np.stack(preds).astype(np.int16) - ref[anchor:].astype(np.int16)).max() <= 1
    assert np.array_equal(drvs[0], U.tensor2img([drv[anchor][None].clone()], rgb2bgr=True, min_max=(-1, 1)))


@pytest.mark.gpu
def test_model_test_entry_and_metrics(tmp_path):
    metrics = {"psnr": {"type": "calculate_psnr", "crop_border": 0, "test_y_channel": False},
               "l1": {"type": "calculate_l1", "crop_border": 0}, "fid": {"type": "calculate_fid"}}
    model = _model(tmp_path, {"relative": False, "adapt_scale": False, "w": 1, "metrics": metrics})
    src, drv = clip(3, 123)
    model.feed_data({"driving": drv[:2], "source": src[None].repeat(2, 1, 1, 1)})
    model.test()
    assert model.out_dict["out"].shape == (2, 3, 256, 256) and model.lq_recon.shape == (2, 3, 256, 256)
    assert set(model.driving_feat) == set(model.source_feat) and set(model.get_current_visuals()) == {"gt", "source", "result", "recon"}     # recon = generator(lq_feat) (appmotioncomp_model.py:590-592)
    item = {"source": src[None], "driving_video": [f[None] for f in drv], "anchor_idx": torch.tensor([0]),
            "video_name": ["c"], "driving_name_list": [[f"{i}"] for i in range(3)]}
    res = model.generate_video_image(FakeLoader([item]), "m", None)
    # reference quirk kept: each array metric is divided by the count over ALL array metrics (2 x frames)
    preds, drvs = model.make_animation(src[None], [f[None] for f in drv])
    from synergize_motion_appearance_amd.models import calculate_psnr, calculate_l1
    assert abs(res["psnr"] - sum(calculate_psnr(p, d) for p, d in zip(preds, drvs)) / 6) < 1e-9
    assert abs(res["l1"] - sum(calculate_l1(p, d) for p, d in zip(preds, drvs)) / 6) < 1e-9
    assert abs(res["l1_255"] - res["l1"] / 255.0) < 1e-12 and np.isnan(res["fid"])
    with pytest.raises(TypeError):
        model.generate_video_image(FakeLoader([dict(item, anchor_idx=None)]), "m", None)
    with pytest.raises(RuntimeError, match="is_train"):               # a test-mode model has no optimiser state (training: test_gpu_train_full)
        model.optimize_parameters(0)


@pytest.mark.gpu
def test_animate_pipeline_from_yml_dataset_and_checkpoints(tmp_path):
    """basicsr/animate.py end to end: pairs csv + PNG folders -> dataset -> loader -> build_model with
    checkpoints on disk ('params_ema' key, one with 'module.' prefixes) -> PNGs; agrees with the
    batched driver run directly on the decoded frames."""
    import basicsr.animate as A
    from synergize_motion_appearance_amd import driver
    from basicsr.archs import build_network
    csv, _, _ = _write_clip(str(tmp_path / "data"), n=5, seed=17)
    cfg = yaml.safe_load(open(os.path.join(REPO, "options/test.yml")))
    ck_g, ck_m = str(tmp_path / "g.pth"), str(tmp_path / "m.pth")
    torch.save({"params_ema": {"module." + k: v for k, v in weights("network_g").items()}}, ck_g)
    torch.save({"params": dict(weights("network_motion_estimator"))}, ck_m)
    cfg["path"] = {"pretrain_network_g": ck_g, "param_key_g": "params_ema", "strict_load_g": True,
                   "pretrain_network_motion_estimator": ck_m, "strict_load_motion_estimator": True,
                   "save_path": str(tmp_path / "log")}
    cfg["val"] = {"relative": True, "adapt_scale": True, "metrics": {"psnr": {"type": "calculate_psnr", "crop_border": 0}}}
    cfg["datasets"] = {"test_1": {"name": "synth_pairs", "type": "FramesMotionTransferTestDataset_CrossID_videopair_anchor",
                                  "root_dir": str(tmp_path / "data"), "pairs_list": csv, "gt_size": 256,
                                  "io_backend": {"type": "disk"}}}
    yml = str(tmp_path / "animate.yml")
    yaml.safe_dump(cfg, open(yml, "w"))
    opt, results = A.test_pipeline(str(tmp_path), argv=["-opt", yml])
    vis = os.path.join(opt["path"]["visualization"], "synth_pairs")
    names = sorted(os.listdir(os.path.join(vis, "result")))
    assert names == [f"id01_0000_id02_{i:04d}.png_r.png" for i in (1, 2, 3, 4)]
    assert np.isfinite(results["synth_pairs"]["psnr"])
    from basicsr.data import build_dataset
    it = build_dataset(dict(cfg["datasets"]["test_1"], phase="test"))[0]
    net_g, me = build_network(cfg["network_g"]).cuda().eval(), build_network(cfg["network_motion_estimator"]).cuda().eval()
    net_g.load_state_dict(weights("network_g"))
    me.load_state_dict(weights("network_motion_estimator"))
    want = driver.animate_batched(it["source"].cuda(), torch.stack(it["driving_video"]).cuda(), net_g, me, relative=True,
                                  adapt_movement_scale=True, batch=4, anchor_idx=1).cpu().numpy()
    got = np.stack([U.imfrombytes(open(os.path.join(vis, "result", n), "rb").read())[:, :, ::-1] for n in names])
    assert np.abs(got.astype(np.int16) - want.astype(np.int16)).max() <= 1


@pytest.mark.gpu
def test_test_py_frame_pair_validation(tmp_path):
    """`basicsr/test.py` end to end (reference test.py:52-80 -> AppMotionCompModel.nondist_validation :463-566): frame-pair csv ->
    FramesMotionTransferTestDataset_PairsList -> model.validation -> result / source / driving / visual PNGs + psnr / l1; the result
    PNG equals the direct module call on the decoded pair (<= 1 LSB) and the visual strip is [source | driving | result | recon]."""
    import importlib.util
    from basicsr.archs import build_network
    from basicsr.data import build_dataset
    csv, _, _ = _write_pairs(str(tmp_path / "data"), n=3, seed=23)
    cfg = yaml.safe_load(open(os.path.join(REPO, "options/test.yml")))
    ck_g, ck_m = str(tmp_path / "g.pth"), str(tmp_path / "m.pth")
    torch.save({"params": dict(weights("network_g"))}, ck_g)
    torch.save({"params": dict(weights("network_motion_estimator"))}, ck_m)
    cfg["path"] = {"pretrain_network_g": ck_g, "param_key_g": "params", "strict_load_g": True, "pretrain_network_motion_estimator": ck_m,
                   "strict_load_motion_estimator": True, "save_path": str(tmp_path / "log")}
    cfg["val"] = {"save_img": True, "metrics": {"psnr": {"type": "calculate_psnr", "crop_border": 0}, "l1": {"type": "calculate_l1"},
                                                "lpips": {"type": "calculate_lpips"}}}
    cfg["datasets"] = {"test_1": {"name": "pairs", "type": "FramesMotionTransferTestDataset_PairsList", "root_dir": str(tmp_path / "data"),
                                  "pairs_list": csv, "gt_size": 256, "io_backend": {"type": "disk"}}}
    yml = str(tmp_path / "test.yml")
    yaml.safe_dump(cfg, open(yml, "w"))
    spec = importlib.util.spec_from_file_location("smx_test_entry", os.path.join(REPO, "basicsr", "test.py"))
    T = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(T)
    opt, results = T.test_pipeline(str(tmp_path), argv=["-opt", yml])
    vis = os.path.join(opt["path"]["visualization"], "pairs")
    for sub, sfx in (("result", "r"), ("source", "s"), ("driving", "d"), ("visual", "v")):
        assert sorted(os.listdir(os.path.join(vis, sub))) == [f"id01_0000_id02_{i:04d}_{sfx}.png" for i in (1, 2)], sub
    m = results["pairs"]
    assert np.isfinite(m["psnr"]) and np.isfinite(m["l1"]) and abs(m["l1_255"] - m["l1"] / 255.0) < 1e-12 and np.isnan(m["lpips"])
    it = build_dataset(dict(cfg["datasets"]["test_1"], phase="test"))[0]
    net_g, me = build_network(cfg["network_g"]).cuda().eval(), build_network(cfg["network_motion_estimator"]).cuda().eval()
    net_g.load_state_dict(weights("network_g"))
    me.load_state_dict(weights("network_motion_estimator"))
    dm = me(it["driving"][None].cuda(), it["source"][None].cuda())
    want = U.tensor2img([net_g(it["source"][None].cuda(), dm, w=1, inference=True)["out"].cpu()], rgb2bgr=True, min_max=(-1, 1))
    got = U.imfrombytes(open(os.path.join(vis, "result", "id01_0000_id02_0001_r.png"), "rb").read())
    assert np.abs(got.astype(np.int16) - want.astype(np.int16)).max() <= 1
    strip = U.imfrombytes(open(os.path.join(vis, "visual", "id01_0000_id02_0001_v.png"), "rb").read())
    assert strip.shape == (256, 4 * 256, 3) and np.array_equal(strip[:, 512:768], got)
