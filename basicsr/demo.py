"""`python basicsr/demo.py --config options/test.yml --source_image S --driving_video D --result_video R [--relative]
[--adapt_scale] [--best_frame i] [--visual_video V]` -- the single-clip inference entry with the reference's command line
(reference `basicsr/demo.py:136-249`), on the MI355X path:

    source image  ->  uint8 H2D, resized / normalised on the device (demo.py:177-185)   ->  source state (encoder taps, kp_source)
    driving clip  ->  driver.FramePipeline: pinned uint8 H2D | device resize+normalise | render B frames | uint8 D2H, three streams
    result        ->  `mimsave` (a video file when imageio is importable, else `<result>.frames/%06d.png`)

What the image this was built in does not have is stated instead of faked: containers are decoded / encoded only when `imageio`
is importable (otherwise the driving clip is a folder of PNG frames), `--find_best_frame` needs `face_alignment`, `--audio`
needs `ffmpeg`, and there is no CPU mode (`--cpu` raises: the product path is the HIP library)."""
import argparse
import os
import sys
from os import path as osp

sys.path.insert(0, osp.abspath(osp.join(osp.dirname(osp.abspath(__file__)), osp.pardir)))

import numpy as np  # noqa: E402
import torch  # noqa: E402
import yaml  # noqa: E402

from basicsr.archs import build_network  # noqa: E402
from basicsr.utils import mimsave  # noqa: E402
from basicsr.utils.options import ordered_yaml  # noqa: E402
from synergize_motion_appearance_amd import driver, ops  # noqa: E402
from synergize_motion_appearance_amd.png import decode_png  # noqa: E402

FRAME_EXT = (".png",)
DEFAULT_FPS = 25.0


def cli(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--config", required=True, help="options yml (network_g, network_motion_estimator, path)")
    ap.add_argument("--source_image", default="source.png")
    ap.add_argument("--driving_video", default="driving.mp4", help="video file (needs imageio) or a folder of PNG frames")
    ap.add_argument("--result_video", default="result.mp4")
    ap.add_argument("--visual_video", default=None, help="also write [source | driving | result] side by side")
    ap.add_argument("--relative", action="store_true", help="relative keypoint motion (vs absolute coordinates)")
    ap.add_argument("--adapt_scale", action="store_true", help="scale the motion by the keypoint hull-area ratio")
    ap.add_argument("--find_best_frame", action="store_true", help="pick the driving frame best aligned with the source (needs face_alignment)")
    ap.add_argument("--best_frame", type=int, default=None, help="anchor frame index: animate backward and forward from it")
    ap.add_argument("--cpu", action="store_true", help="(not supported: the product path is the HIP library)")
    ap.add_argument("--audio", action="store_true", help="copy the driving video's audio track (needs ffmpeg)")
    ap.add_argument("--batch", type=int, default=60, help="driving frames in flight per step")
    return ap.parse_args(argv)


def read_rgb(path):
    """image file -> uint8 RGB [H,W,3] (PNG through the in-tree decoder; anything else needs imageio)."""
    if path.lower().endswith(FRAME_EXT):
        with open(path, "rb") as f:
            img = decode_png(f.read())
        img = np.repeat(img[:, :, None], 3, 2) if img.ndim == 2 else img[..., :3]
        return np.ascontiguousarray(img)
    try:
        import imageio
    except ImportError as e:
        raise RuntimeError(f"{path}: only PNG images are readable without imageio") from e
    return np.ascontiguousarray(np.asarray(imageio.imread(path))[..., :3])


def read_clip(path):
    """driving clip -> (uint8 RGB [N,H,W,3], fps): a folder of PNG frames in name order, or a container via imageio."""
    if osp.isdir(path):
        names = sorted(n for n in os.listdir(path) if n.lower().endswith(FRAME_EXT))
        if not names:
            raise RuntimeError(f"{path}: no PNG frames")
        from synergize_motion_appearance_amd.png import pool
        frames = list(pool().map(read_rgb, [osp.join(path, n) for n in names]))     # decoded on the codec thread pool
        if len({f.shape for f in frames}) != 1:
            raise RuntimeError(f"{path}: frames differ in size")
        return np.stack(frames), DEFAULT_FPS
    try:
        import imageio
    except ImportError as e:
        raise RuntimeError(f"{path}: decoding a video container needs imageio; pass a folder of PNG frames instead") from e
    reader = imageio.get_reader(path)
    fps = reader.get_meta_data().get("fps", DEFAULT_FPS)
    frames = []
    try:
        for im in reader:
            frames.append(np.asarray(im)[..., :3])
    except RuntimeError:                                     # truncated streams end this way (reference demo.py:169-173)
        pass
    reader.close()
    return np.stack(frames), fps


def clip_loaders(path):
    """a folder of PNG frames -> (zero-argument loaders in name order, frame (H, W)): the frames are decoded later, on the codec pool."""
    names = sorted(n for n in os.listdir(path) if n.lower().endswith(FRAME_EXT))
    if not names:
        raise RuntimeError(f"{path}: no PNG frames")
    first = read_rgb(osp.join(path, names[0]))
    return [(lambda q=osp.join(path, n): read_rgb(q)) for n in names], first.shape[:2]


@torch.no_grad()
def animate_folder(net_g, me, source_rgb, frames_dir, out_dir, relative, adapt_scale, anchor=0, batch=60, level=1, pipe=None):
    """folder of PNG driving frames -> folder of PNG result frames, streaming: the frames are decoded on the codec thread pool ahead of the GPU
    (driver.LazyFrames -> pinned staging buffers), rendered in batches, and every finished batch is handed back to the pool for encoding while
    the next one renders (reference demo.py:166-185 reads the whole clip, :103-134 loops, :222 writes it).  `pipe`: a FramePipeline to reuse (a server
    animating clip after clip keeps its captured hipGraph and staging buffers).  -> number of frames written"""
    from synergize_motion_appearance_amd.png import pool, write_png
    dev = next(net_g.parameters()).device
    loaders, hw = clip_loaders(frames_dir)
    lazy = driver.LazyFrames(loaders, hw)
    lazy.prefetch()
    src = ops.frames_u8_to_nchw(torch.from_numpy(source_rgb)[None].to(dev), (256, 256))
    first = ops.frames_u8_to_nchw(torch.from_numpy(np.ascontiguousarray(lazy.frame(anchor)))[None].to(dev), (256, 256)) if (relative or adapt_scale) else None
    state = driver.encode_source_state(net_g, me, src, first, adapt_scale)
    if pipe is None:
        pipe = driver.FramePipeline(net_g, me, batch=min(batch, len(lazy)), frame_hw=hw, relative=relative, adapt_movement_scale=adapt_scale)
    os.makedirs(out_dir, exist_ok=True)
    futs = []
    for a, chunk in pipe.stream(state, lazy):
        arr = chunk.numpy().copy()                              # the pinned buffer is recycled two batches later
        futs += [pool().submit(write_png, osp.join(out_dir, f"{a + i:06d}.png"), arr[i], level) for i in range(arr.shape[0])]
    for f in futs:
        f.result()
    return len(futs)


def load_checkpoint(net, path, strict=True, key="params"):
    """`torch.load(path)[key]` with DataParallel's `module.` prefix dropped, then a (strict) load (reference demo.py:58-71)."""
    blob = torch.load(path, map_location="cpu")
    if key is not None:
        blob = blob[key if key in blob else "params"]
    net.load_state_dict({(k[7:] if k.startswith("module.") else k): v for k, v in blob.items()}, strict=strict)
    return net


def build_from_config(config, device):
    p = config.get("path", {}) or {}
    net_g = build_network(config["network_g"])
    if p.get("pretrain_network_g"):
        load_checkpoint(net_g, p["pretrain_network_g"], p.get("strict_load_g", True), p.get("param_key_g", "params"))
    me = build_network(config["network_motion_estimator"])
    if p.get("pretrain_network_motion_estimator"):
        load_checkpoint(me, p["pretrain_network_motion_estimator"], p.get("strict_load_motion_estimator", True), p.get("param_key_m", "params"))
    return net_g.to(device).eval(), me.to(device).eval()


@torch.no_grad()
def animate(net_g, me, source_rgb, clip_rgb, relative, adapt_scale, anchor=0, batch=60):
    """uint8 RGB source [H,W,3] + clip [N,h,w,3] -> uint8 RGB frames [N,256,256,3] (host).  `anchor`: the frame whose keypoints
    are the initial ones -- animating backward and forward from it and splicing (demo.py:205-216) equals this single pass."""
    dev = next(net_g.parameters()).device
    src = ops.frames_u8_to_nchw(torch.from_numpy(source_rgb)[None].to(dev), (256, 256))
    first = ops.frames_u8_to_nchw(torch.from_numpy(clip_rgb[anchor:anchor + 1]).to(dev), (256, 256)) if (relative or adapt_scale) else None
    state = driver.encode_source_state(net_g, me, src, first, adapt_scale)
    pipe = driver.FramePipeline(net_g, me, batch=min(batch, len(clip_rgb)), frame_hw=clip_rgb.shape[1:3], relative=relative, adapt_movement_scale=adapt_scale)
    return pipe.run(state, torch.from_numpy(clip_rgb)).numpy(), ops.to_uint8(src.permute(0, 2, 3, 1).contiguous(), -1.0, 1.0)[0].cpu().numpy()


def main(argv=None):
    opt = cli(argv)
    if opt.cpu:
        raise SystemExit("--cpu: there is no CPU mode; the path runs on the HIP library (libsmx.so) on an MI355X")
    if opt.find_best_frame and opt.best_frame is None:
        raise SystemExit("--find_best_frame needs the face_alignment package (not available here); pass --best_frame <index>")
    if opt.audio:
        print("--audio: no ffmpeg here; the result is written without an audio track", file=sys.stderr)
    with open(opt.config) as f:
        config = yaml.load(f, Loader=ordered_yaml()[0])
    if not torch.cuda.is_available():
        raise SystemExit("demo.py needs an MI355X (no CPU fallback for the HIP path)")
    dev = torch.device("cuda", torch.cuda.current_device())
    net_g, me = build_from_config(config, dev)
    source = read_rgb(opt.source_image)
    try:
        import imageio  # noqa: F401
        have_imageio = True
    except ImportError:
        have_imageio = False
    if osp.isdir(opt.driving_video) and not have_imageio and opt.visual_video is None:
        # PNG folder in, PNG folder out: the streaming path (decode | H2D | render | D2H | encode overlapped)
        anchor = 0 if opt.best_frame is None else int(opt.best_frame)
        out_dir = str(opt.result_video) + ".frames"
        n = animate_folder(net_g, me, source, opt.driving_video, out_dir, opt.relative, opt.adapt_scale, anchor, opt.batch)
        print(f"{n} frames -> {out_dir}")
        return None
    clip, fps = read_clip(opt.driving_video)
    anchor = 0 if opt.best_frame is None else int(opt.best_frame)
    if not 0 <= anchor < len(clip):
        raise SystemExit(f"--best_frame {anchor}: the clip has {len(clip)} frames")
    if opt.best_frame is not None:
        print(f"Best frame: {anchor}")
    frames, source256 = animate(net_g, me, source, clip, opt.relative, opt.adapt_scale, anchor, opt.batch)
    where = mimsave(list(frames), opt.result_video, fps=fps)
    print(f"{len(frames)} frames -> {where or opt.result_video}")
    if opt.visual_video is not None:
        drv256 = ops.to_uint8(ops.frames_u8_to_nchw(torch.from_numpy(clip).to(dev), (256, 256)).permute(0, 2, 3, 1).contiguous(), -1.0, 1.0).cpu().numpy()
        vis = [np.concatenate((source256, d, p), axis=1) for d, p in zip(drv256, frames)]
        mimsave(vis, opt.visual_video, fps=fps)
    return frames


if __name__ == "__main__":
    main()
