"""Batched reenactment driver -- the MI355X-native counterpart of the reference's per-frame
loop `make_animation` / `normalize_kp` (`basicsr/demo.py:24-44,103-134`).

Given (kp_source, kp_driving_initial, adapt scale) every driving frame is independent
(no recurrent state), so frames are processed `batch` at a time with the source encoding
cached, and -- across GPUs -- sharded in contiguous blocks with one RCCL broadcast of the
source cache (`shard_frames`, `broadcast_source_cache`).
"""
import math

import numpy as np
import torch

from . import ops


def hull_area(points) -> float:
    """scipy ConvexHull 'volume' (= area in 2-D) of kp values [K,2], fp64 on the host (demo.py:27-28)."""
    from scipy.spatial import ConvexHull
    return float(ConvexHull(np.asarray(points, dtype=np.float64)).volume)


def adapt_scale(kp_source, kp_driving_initial) -> float:
    a = hull_area(kp_source["value"][0].detach().cpu().numpy())
    b = hull_area(kp_driving_initial["value"][0].detach().cpu().numpy())
    return math.sqrt(a) / math.sqrt(b)


def normalize_kp(kp_source, kp_driving, kp_driving_initial, adapt_movement_scale=False,
                 use_relative_movement=False, use_relative_jacobian=False, scale=None):
    """demo.py:24-44. `scale` lets the caller pass the (frame-invariant) hull ratio computed once."""
    if adapt_movement_scale:
        s = adapt_scale(kp_source, kp_driving_initial) if scale is None else scale
    else:
        s = 1
    if use_relative_movement and kp_driving["value"].is_cuda and kp_source["value"].shape[0] == 1:
        # device tensors: one small HIP kernel for the whole batch (no ATen matmul / inverse on the path)
        return ops.normalize_kp(kp_driving, kp_driving_initial, kp_source, s, True, use_relative_jacobian)
    kp_new = {k: v for k, v in kp_driving.items()}
    if use_relative_movement:
        kp_new["value"] = (kp_driving["value"] - kp_driving_initial["value"]) * s + kp_source["value"]
        if use_relative_jacobian:
            jd = torch.matmul(kp_driving["jacobian"], torch.inverse(kp_driving_initial["jacobian"]))
            kp_new["jacobian"] = torch.matmul(jd, kp_source["jacobian"])
    return kp_new


def shard_frames(n_frames: int, rank: int, world: int):
    """contiguous block of frame indices owned by `rank` (37/38 per GPU for 300 frames on 8)."""
    base, extra = divmod(n_frames, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


CACHE_SHAPES = {32: (1, 32, 32, 256), 64: (1, 64, 64, 128), 128: (1, 128, 128, 128), 256: (1, 256, 256, 64)}
CACHE_ORDER = (32, 64, 128, 256)
_KP_VALUE, _KP_JAC = 15 * 2, 15 * 4


def cache_numel() -> int:
    n = _KP_VALUE + _KP_JAC
    for s in CACHE_ORDER:
        c = 1
        for d in CACHE_SHAPES[s]:
            c *= d
        n += c
    return n          # 7,077,888 encoder elements (28.3 MB fp32) + 90 keypoint floats


def pack_source_cache(cache, kp_source) -> torch.Tensor:
    """flatten the frame-invariant state of one source into ONE buffer (one collective, not six)."""
    return torch.cat([cache.feats[s].reshape(-1) for s in CACHE_ORDER] +
                     [kp_source["value"].reshape(-1), kp_source["jacobian"].reshape(-1)])


def unpack_source_cache(flat: torch.Tensor):
    from .engine_netg import SourceCache
    feats, off = {}, 0
    for s in CACHE_ORDER:
        n = 1
        for d in CACHE_SHAPES[s]:
            n *= d
        feats[s] = flat[off:off + n].view(CACHE_SHAPES[s])
        off += n
    kp = {"value": flat[off:off + _KP_VALUE].view(1, 15, 2),
          "jacobian": flat[off + _KP_VALUE:off + _KP_VALUE + _KP_JAC].view(1, 15, 2, 2)}
    return SourceCache(feats, 1), kp


def broadcast_flat(flat_or_none, device, src=0):
    """rank `src` passes the packed cache, the others None; everyone returns the broadcast buffer.
    torch.distributed broadcast == RCCL over xGMI on the GPU box (gloo in the CPU tests)."""
    import torch.distributed as dist
    buf = flat_or_none if dist.get_rank() == src else torch.empty(cache_numel(), device=device, dtype=torch.float32)
    dist.broadcast(buf, src=src)
    return buf


def broadcast_source_cache(net_g, motion_estimator, source, src=0):
    """rank `src` encodes the source once; every rank receives the frame-invariant cache
    (encoder taps 28.3 MB fp32 + kp_source) with ONE broadcast."""
    import torch.distributed as dist
    flat = None
    if dist.get_rank() == src:
        cache = net_g.engine().encode_source(source.float())
        flat = pack_source_cache(cache, motion_estimator.engine().estimate_kp(source.float()))
    return unpack_source_cache(broadcast_flat(flat, source.device, src))


@torch.no_grad()
def animate_batched(source, driving, net_g, motion_estimator, relative=True, adapt_movement_scale=True,
                    batch=8, kp_source=None, kp_driving_initial=None, source_cache=None, want="uint8", anchor_idx=0, w=1.0):
    """source [3,H,W] / [1,3,H,W], driving [N,3,H,W] device tensors in [-1,1].
    -> uint8 frames [N,H,W,3] (want='uint8'), fp32 NCHW [N,3,H,W] ('float'), or both ('both').
    anchor_idx: the frame whose keypoints are `kp_driving_initial`.  The reference's dataset path
    (models/appmotioncomp_model.py:675-683) animates forward from the anchor and backward from it
    and splices the two lists; both runs use kp(driving[anchor]) as the initial keypoints and frames
    are otherwise independent, so that equals ONE pass over all frames with this anchor."""
    src = source if source.dim() == 4 else source.unsqueeze(0)
    eng_g, eng_m = net_g.engine(), motion_estimator.engine()
    if kp_source is None:
        kp_source = eng_m.estimate_kp(src.float())
    if kp_driving_initial is None and (relative or adapt_movement_scale):
        kp_driving_initial = eng_m.estimate_kp(driving[anchor_idx:anchor_idx + 1].float())
    scale = adapt_scale(kp_source, kp_driving_initial) if adapt_movement_scale else None
    cache = eng_g.encode_source(src.float()) if source_cache is None else source_cache
    src64 = eng_m.source_down(src.float())
    u8, fl = [], []
    for i in range(0, driving.shape[0], batch):
        frames = driving[i:i + batch].float()
        kp_d = eng_m.estimate_kp(frames)
        kp_n = normalize_kp(kp_source, kp_d, kp_driving_initial, adapt_movement_scale, relative, relative, scale)
        dm = eng_m.dense_motion(src64, kp_n, kp_source)
        st = eng_g.forward(cache, dm["deformation"], dm["occlusion_nhwc"].view(-1, 64, 64), dm["heat_nhwc"], float(w))
        if want in ("uint8", "both"):
            u8.append(ops.to_uint8(st["out"], -1.0, 1.0))
        if want in ("float", "both"):
            fl.append(ops.nhwc_to_nchw(st["out"]))
    r8 = torch.cat(u8) if u8 else None
    rf = torch.cat(fl) if fl else None
    return r8 if want == "uint8" else rf if want == "float" else (r8, rf)


@torch.no_grad()
def make_animation(source_image, driving_video, net_g, motion_estimator, relative=True,
                   adapt_movement_scale=True, cpu=False, batch=8):
    """demo.py:103-134 signature; returns (predictions, driving_imgs) as lists of uint8 HWC arrays."""
    if cpu:
        raise RuntimeError("the MI355X-native path has no CPU mode")
    drv = torch.stack(list(driving_video)) if not torch.is_tensor(driving_video) else driving_video
    dev = next(net_g.parameters()).device
    out = animate_batched(source_image.to(dev), drv.to(dev), net_g, motion_estimator, relative, adapt_movement_scale, batch)
    preds = list(out.cpu().numpy())
    drv8 = ops.to_uint8(drv.to(dev).permute(0, 2, 3, 1).contiguous(), -1.0, 1.0).cpu().numpy()
    return preds, list(drv8)
