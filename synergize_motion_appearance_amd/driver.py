"""Batched reenactment driver -- the MI355X-native counterpart of the reference's per-frame
loop `make_animation` / `normalize_kp` (`basicsr/demo.py:24-44,103-134`).

Given (kp_source, kp_driving_initial, adapt scale) every driving frame is independent
(no recurrent state), so frames are processed `batch` at a time with the source encoding
cached, and -- across GPUs -- sharded in contiguous blocks with one RCCL broadcast of the
source state (`shard_frames`, `broadcast_source_state`, `animate_sharded`).
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
    kp_new = {k: v for k, v in kp_driving.items()}       # every key of kp_driving survives (demo.py:34)
    fast = (use_relative_movement and kp_driving["value"].is_cuda and kp_source["value"].shape[0] == 1
            and kp_driving_initial["value"].shape[0] == 1)
    if fast:
        # device tensors, ONE source and ONE initial frame (the kernel indexes row 0 of both): one small HIP kernel
        # for the whole batch (no ATen matmul / inverse on the path); batched initial keypoints take the torch path.
        # `s` may be a one-float device tensor (a broadcast state's hull ratio, NaN = none): read on the device
        kp_new.update(ops.normalize_kp(kp_driving, kp_driving_initial, kp_source, s, True, use_relative_jacobian))
        return kp_new
    if torch.is_tensor(s):
        s = torch.where(torch.isnan(s), torch.ones_like(s), s)          # NaN encodes "none" (pack_source_state)
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
_SRC64 = 64 * 64 * 3
# the frame-invariant state of one (source, clip): encoder taps | down(source) 64x64x3 | kp_source | kp_driving_initial | adapt scale
_TAIL = _SRC64 + 2 * (_KP_VALUE + _KP_JAC) + 1


def _numel(shape):
    n = 1
    for d in shape:
        n *= d
    return n


def _img_size(net_g) -> int:
    """frame size of the generator network (options yml `network_g.img_size`): 256, or 512 for the N4 variant."""
    return int(getattr(net_g, "cfg", {}).get("img_size", 256))


def cache_shapes(img_size=256):
    """encoder-tap shapes keyed by the scale's name (its size in the 256 layout); every grid scales with img_size / 256 (N4)."""
    k = img_size // 256
    return {s: (1, sh[1] * k, sh[2] * k, sh[3]) for s, sh in CACHE_SHAPES.items()}


def cache_numel(dtype=torch.float32, img_size=256) -> int:
    """floats in the packed state: 7,077,888 encoder elements (28.3 MB fp32; bf16 taps travel as raw bits, two per
    float: 14.2 MB) + 12,288 (down-sampled source) + 2 x 90 keypoint floats + 1 at img_size 256; the encoder and
    down-sampled-source parts x4 at 512."""
    k = img_size // 256
    n = sum(_numel(CACHE_SHAPES[s]) for s in CACHE_ORDER) * k * k
    return (n // 2 if dtype == torch.bfloat16 else n) + _SRC64 * k * k + _TAIL - _SRC64


class SourceState:
    """everything a rank needs to render frames of one (source, clip): the SourceCache, src64, kp_source,
    kp_driving_initial and the adapt-movement scale.  `flat` is the packed buffer it was unpacked from (views)."""
    __slots__ = ("cache", "src64", "kp_source", "kp_initial", "scale", "flat")

    def __init__(self, cache, src64, kp_source, kp_initial, scale, flat=None):
        self.cache, self.src64, self.kp_source, self.kp_initial, self.scale, self.flat = cache, src64, kp_source, kp_initial, scale, flat


def pack_source_state(cache, src64, kp_source, kp_initial, scale) -> torch.Tensor:
    """flatten the frame-invariant state into ONE buffer (one collective, not ten)."""
    dev = src64.device
    kp0 = kp_initial if kp_initial is not None else {"value": torch.zeros((1, 15, 2), device=dev), "jacobian": torch.zeros((1, 15, 2, 2), device=dev)}
    bits = lambda t: t.reshape(-1).view(torch.float32) if t.dtype == torch.bfloat16 else t.reshape(-1)   # noqa: E731  (bf16 taps: raw bits)
    return torch.cat([bits(cache.feats[s]) for s in CACHE_ORDER] +
                     [src64.reshape(-1), kp_source["value"].reshape(-1), kp_source["jacobian"].reshape(-1),
                      kp0["value"].reshape(-1), kp0["jacobian"].reshape(-1),
                      torch.full((1,), float("nan") if scale is None else float(scale), device=dev, dtype=torch.float32)])


def unpack_source_state(flat: torch.Tensor, dtype=torch.float32, img_size=256) -> SourceState:
    """dtype: storage type of the encoder taps inside `flat` (the receiving engine's activation type)."""
    from .engine_netg import SourceCache
    feats, off = {}, 0
    shapes, k = cache_shapes(img_size), img_size // 256
    if flat.numel() != cache_numel(dtype, img_size):
        raise ValueError(f"unpack_source_state: {flat.numel()} floats, a packed img_size-{img_size} state has {cache_numel(dtype, img_size)}")
    for s in CACHE_ORDER:
        n = _numel(shapes[s])
        if dtype == torch.bfloat16:
            feats[s] = flat[off:off + n // 2].view(torch.bfloat16).view(shapes[s])
            off += n // 2
        else:
            feats[s] = flat[off:off + n].view(shapes[s])
            off += n
    src64 = flat[off:off + _SRC64 * k * k].view(1, 64 * k, 64 * k, 3)
    off += _SRC64 * k * k
    kps = []
    for _ in range(2):
        kps.append({"value": flat[off:off + _KP_VALUE].view(1, 15, 2),
                    "jacobian": flat[off + _KP_VALUE:off + _KP_VALUE + _KP_JAC].view(1, 15, 2, 2)})
        off += _KP_VALUE + _KP_JAC
    # the hull ratio stays where it arrived (a one-float view of the packed buffer, NaN = none): the normalize_kp kernel reads it
    # from device memory, so unpacking costs no host synchronisation and N broadcasts can be in flight at once
    return SourceState(SourceCache(feats, 1), src64, kps[0], kps[1], flat[off:off + 1], flat)


def broadcast_flat(flat_or_none, device, src=0, group=None, dtype=torch.float32, async_op=False, img_size=256):
    """rank `src` (a GLOBAL rank, as torch.distributed.broadcast reads it) passes the packed state, the others None;
    everyone returns the broadcast buffer -- or (buffer, work handle) with async_op=True.
    torch.distributed broadcast == RCCL over xGMI on the GPU box (gloo in the CPU tests)."""
    import torch.distributed as dist
    buf = flat_or_none if dist.get_rank() == src else torch.empty(cache_numel(dtype, img_size), device=device, dtype=torch.float32)
    if dist.get_rank() == src and buf is None:
        raise ValueError(f"broadcast_flat: rank {src} is the source of this broadcast and must pass the packed state")
    work = dist.broadcast(buf, src=src, group=group, async_op=async_op)
    return (buf, work) if async_op else buf


def broadcast_source_states(net_g, motion_estimator, owned, owners, adapt_movement_scale=True, device=None, group=None):
    """the frame-invariant state of SEVERAL sources, each encoded by its owner rank: `owners[j]` = global owner rank of source j,
    `owned[j]` = (source, initial_frame) on the owner.  Every rank first encodes ALL the sources it owns, then all the
    broadcasts are issued back to back (async) and waited for together -- the encodes of different owners overlap each
    other and the N transfers pipeline, instead of N x (encode -> broadcast -> host sync) in sequence.  -> {j: SourceState}"""
    import torch.distributed as dist
    me = dist.get_rank()
    adt = net_g.engine().adt
    flats = {}
    for j, owner in owners.items():
        if owner == me:
            st = encode_source_state(net_g, motion_estimator, owned[j][0], owned[j][1], adapt_movement_scale)
            flats[j] = pack_source_state(st.cache, st.src64, st.kp_source, st.kp_initial, st.scale)
            device = flats[j].device
    if device is None:
        device = next(net_g.parameters()).device
    img = _img_size(net_g)
    pending = [(j,) + broadcast_flat(flats.get(j), device, owner, group, adt, async_op=True, img_size=img) for j, owner in sorted(owners.items())]
    out = {}
    for j, buf, work in pending:
        work.wait()
        out[j] = unpack_source_state(buf, adt, img)
    return out


def encode_source_state(net_g, motion_estimator, source, initial_frame=None, adapt_movement_scale=True) -> SourceState:
    """the frame-invariant work of one (source, clip) on THIS rank: source encoder taps (A8), down(source), kp_source,
    kp(initial driving frame) and the hull-area scale (demo.py:24-32, 114-121)."""
    eng_g, eng_m = net_g.engine(), motion_estimator.engine()
    src = source if source.dim() == 4 else source.unsqueeze(0)
    src = src.float()
    kp_s = eng_m.estimate_kp(src)
    kp_0 = None if initial_frame is None else eng_m.estimate_kp((initial_frame if initial_frame.dim() == 4 else initial_frame.unsqueeze(0)).float())
    scale = adapt_scale(kp_s, kp_0) if (adapt_movement_scale and kp_0 is not None) else None
    return SourceState(eng_g.encode_source(src), eng_m.source_down(src), kp_s, kp_0, scale)


def broadcast_source_state(net_g, motion_estimator, source=None, initial_frame=None, adapt_movement_scale=True,
                           src=0, device=None, group=None) -> SourceState:
    """rank `src` (GLOBAL rank) encodes the source once (`source` / `initial_frame` are only read there); every rank receives
    the frame-invariant state with ONE broadcast of 28.4 MB -- the only collective on the data path."""
    import torch.distributed as dist
    flat = None
    if dist.get_rank() == src:
        st = encode_source_state(net_g, motion_estimator, source, initial_frame, adapt_movement_scale)
        flat = pack_source_state(st.cache, st.src64, st.kp_source, st.kp_initial, st.scale)
        device = flat.device
    if device is None:
        device = next(net_g.parameters()).device
    adt = net_g.engine().adt
    img = _img_size(net_g)
    return unpack_source_state(broadcast_flat(flat, device, src, group, adt, img_size=img), adt, img)


def render_frames(state: SourceState, frames, net_g, motion_estimator, relative=True, adapt_movement_scale=True,
                  batch=30, want="uint8", w=1.0):
    """frames [n,3,H,W] (any subset of a clip, in any order: frames are independent given `state`) -> uint8 [n,H,W,3]
    / fp32 NCHW [n,3,H,W] / both."""
    eng_g, eng_m = net_g.engine(), motion_estimator.engine()
    u8, fl = [], []
    if frames.shape[0] == 0:       # an empty shard (fewer frames than ranks): typed empty results, so collectives still line up
        H, W = frames.shape[-2:]
        r8 = torch.empty((0, H, W, 3), device=frames.device, dtype=torch.uint8)
        rf = torch.empty((0, 3, H, W), device=frames.device, dtype=torch.float32)
        return r8 if want == "uint8" else rf if want == "float" else (r8, rf)
    for i in range(0, frames.shape[0], batch):
        kp_d = eng_m.estimate_kp(frames[i:i + batch].float())
        kp_n = normalize_kp(state.kp_source, kp_d, state.kp_initial, adapt_movement_scale, relative, relative, state.scale)
        dm = eng_m.dense_motion(state.src64, kp_n, state.kp_source)
        st = eng_g.forward(state.cache, dm["deformation"], dm["occlusion_nhwc"], dm["heat_nhwc"], float(w))
        if want in ("uint8", "both"):
            u8.append(ops.to_uint8(st["out"], -1.0, 1.0))
        if want in ("float", "both"):
            fl.append(ops.nhwc_to_nchw(st["out"]))
    r8 = torch.cat(u8) if u8 else None
    rf = torch.cat(fl) if fl else None
    return r8 if want == "uint8" else rf if want == "float" else (r8, rf)


@torch.no_grad()
def animate_sharded(source, driving, net_g, motion_estimator, relative=True, adapt_movement_scale=True, batch=30,
                    root=0, anchor_idx=0, gather=True, group=None):
    """The N>1 form of `animate_batched` (one process per GPU, torch.distributed initialised): rank `root` encodes the
    source (+ the anchor frame's keypoints and the hull scale) and broadcasts the packed state once; every rank
    renders its contiguous `shard_frames` block of `driving` [N,3,H,W] -- no other collective on the data path.
    gather=True: the uint8 frames are collected on `root` in clip order (returns [N,H,W,3] there, None elsewhere);
    gather=False: returns ((start, stop), frames_of_this_rank).
    `root` is a rank OF `group` (group-local, like `rank` below); it is converted once to the global rank that
    torch.distributed's broadcast / gather expect, so a strict subgroup (e.g. global ranks 4-7, root=0 -> global 4) works."""
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    if rank < 0:
        raise ValueError("animate_sharded: this process is not a member of `group`")
    groot = root if group is None else dist.get_global_rank(group, root)
    dev = driving.device
    need0 = relative or adapt_movement_scale
    state = broadcast_source_state(net_g, motion_estimator, source if rank == root else None,
                                   driving[anchor_idx:anchor_idx + 1] if (rank == root and need0) else None,
                                   adapt_movement_scale, src=groot, device=dev, group=group)
    n = driving.shape[0]
    a, b = shard_frames(n, rank, world)
    mine = render_frames(state, driving[a:b], net_g, motion_estimator, relative, adapt_movement_scale, batch)
    if not gather:
        return (a, b), mine
    sizes = [shard_frames(n, r, world)[1] - shard_frames(n, r, world)[0] for r in range(world)]
    pad = torch.zeros((max(sizes),) + tuple(mine.shape[1:]), device=dev, dtype=mine.dtype)
    pad[:mine.shape[0]] = mine
    if dist.get_backend(group) == "gloo":            # gloo has no device gather: stage the uint8 frames through the host
        pad = pad.cpu()
    parts = [torch.empty_like(pad) for _ in range(world)] if rank == root else None
    dist.gather(pad, parts, dst=groot, group=group)
    return torch.cat([p[:k] for p, k in zip(parts, sizes)]).to(dev) if rank == root else None


@torch.no_grad()
def animate_batched(source, driving, net_g, motion_estimator, relative=True, adapt_movement_scale=True,
                    batch=30, kp_source=None, kp_driving_initial=None, source_cache=None, want="uint8", anchor_idx=0, w=1.0):
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
    state = SourceState(cache, eng_m.source_down(src.float()), kp_source, kp_driving_initial, scale)
    return render_frames(state, driving, net_g, motion_estimator, relative, adapt_movement_scale, batch, want, w)


@torch.no_grad()
def make_animation(source_image, driving_video, net_g, motion_estimator, relative=True,
                   adapt_movement_scale=True, cpu=False, batch=30):
    """demo.py:103-134 signature; returns (predictions, driving_imgs) as lists of uint8 HWC arrays."""
    if cpu:
        raise RuntimeError("the MI355X-native path has no CPU mode")
    drv = torch.stack(list(driving_video)) if not torch.is_tensor(driving_video) else driving_video
    dev = next(net_g.parameters()).device
    out = animate_batched(source_image.to(dev), drv.to(dev), net_g, motion_estimator, relative, adapt_movement_scale, batch)
    preds = list(out.cpu().numpy())
    drv8 = ops.to_uint8(drv.to(dev).permute(0, 2, 3, 1).contiguous(), -1.0, 1.0).cpu().numpy()
    return preds, list(drv8)


def _host_copy(dst, src):
    """CPU tensor -> CPU tensor as ONE plain memcpy.  `Tensor.copy_` between host tensors goes through ATen's parallel copy kernel: its
    OpenMP team (one thread per host CPU) spins after the region, and on a CPU-quota'd container that spinning throttles the very thread
    that issues the next batch's kernel launches -- measured: a 0.8 MB copy per batch made the following ~450 launches take 32 ms instead of
    8 (tools/pipe_prof.py)."""
    if dst.is_contiguous() and src.is_contiguous() and dst.dtype == src.dtype and not dst.is_cuda and not src.is_cuda:
        import numpy as np
        np.copyto(dst.numpy(), src.numpy())
    else:
        dst.copy_(src)


class LazyFrames:
    """A driving clip that is still being decoded: `loaders` are zero-argument callables returning uint8 RGB [H,W,3] arrays (e.g. PNG files of a folder);
    they run on the codec thread pool (png.pool()) a bounded number of batches ahead of the consumer, and `copy_into` fills a (pinned) staging buffer
    with frames [a, a + n) as FramePipeline.stream asks for them -- decode, H2D copy, rendering and the D2H copy of different batches overlap."""

    def __init__(self, loaders, frame_hw, ahead=192):
        from .png import pool
        self.loaders, self.hw, self.ahead = list(loaders), tuple(frame_hw), int(ahead)
        self.shape = (len(self.loaders), self.hw[0], self.hw[1], 3)
        self._pool, self._futs, self._next = pool(), {}, 0

    def __len__(self):
        return len(self.loaders)

    def _submit_to(self, upto):
        upto = min(upto, len(self.loaders))
        while self._next < upto:
            self._futs[self._next] = self._pool.submit(self.loaders[self._next])
            self._next += 1

    def prefetch(self, n=None):
        self._submit_to(self.ahead if n is None else n)

    def frame(self, i):
        self._submit_to(i + 1)
        f = self._futs[i].result()
        img = np.asarray(f)
        img = np.repeat(img[:, :, None], 3, 2) if img.ndim == 2 else img[..., :3]
        if tuple(img.shape[:2]) != self.hw:
            raise ValueError(f"frame {i} is {img.shape[:2]}, the clip is {self.hw}")
        return img

    def copy_into(self, dst, a, n):
        self._submit_to(a + n + self.ahead)
        d = dst.numpy()
        for i in range(n):
            d[i] = self.frame(a + i)
            self._futs.pop(a + i, None)


class FramePipeline:
    """N3 -- the host I/O around the loop (demo.py:166-185 in, :222 out) as an MI355X pipeline: uint8 driving frames leave
    pinned host memory one byte per sample, are resized / normalised ON the device, rendered in batches, and the uint8 result
    frames return to pinned host memory, with the H2D copy of batch i+1 and the D2H copy of batch i-1 overlapping the
    compute of batch i on three HIP streams (double-buffered staging on both sides).  `run` accepts any CPU uint8 tensor /
    numpy array [N,H,W,3] (or an iterable of such chunks) and returns / yields uint8 [n,256,256,3] host tensors.

    use_graph (default on): the per-batch launch sequence (uint8 -> fp32 normalise, keypoints, dense motion, generator, uint8 pack: ~700
    launches, ~8 ms of Python + ctypes) is captured ONCE in a hipGraph over static input / output / source-state buffers and replayed per
    full batch: it matters where a batch's GPU work is shorter than that (small batches in bf16).  A new source is a copy of its packed
    state into the static buffer (the unpacked tensors are views of it); a ragged last batch runs eagerly."""

    def __init__(self, net_g, motion_estimator, batch=60, frame_hw=(256, 256), swap_rb=False, relative=True, adapt_movement_scale=True,
                 use_graph=True):
        self.net_g, self.me, self.B = net_g, motion_estimator, int(batch)
        self.hw, self.swap_rb, self.relative, self.adapt = tuple(frame_hw), swap_rb, relative, adapt_movement_scale
        dev = next(net_g.parameters()).device
        if dev.type != "cuda":
            raise ops.L.SmxError("FramePipeline: the networks must be on an MI355X (call .cuda())")
        self.dev = dev
        H, W = self.hw
        self.img = _img_size(net_g)                  # the network's frame size: 256 (512: DESIGN N4)
        self.pin_in = [torch.empty((self.B, H, W, 3), dtype=torch.uint8).pin_memory() for _ in range(2)]
        self.dev_in = [torch.empty((self.B, H, W, 3), dtype=torch.uint8, device=dev) for _ in range(2)]
        self.pin_out = [torch.empty((self.B, self.img, self.img, 3), dtype=torch.uint8).pin_memory() for _ in range(2)]
        self.dev_out = [torch.empty((self.B, self.img, self.img, 3), dtype=torch.uint8, device=dev) for _ in range(2)]
        self.s_h2d, self.s_d2h = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
        self.use_graph = bool(use_graph)
        self._graph = None           # (CUDAGraph, static uint8 input, static uint8 output, static packed state, its unpacked SourceState, dtype)
        self._graph_src = None       # the SourceState object whose values the static state currently holds

    def _render_eager(self, state, u8):
        x = ops.frames_u8_to_nchw(u8, (self.img, self.img), self.swap_rb)
        return render_frames(state, x, self.net_g, self.me, self.relative, self.adapt, batch=self.B)

    def _render(self, state, u8):
        """uint8 device frames [n,H,W,3] -> uint8 device frames [n,img,img,3]; full batches through the captured graph."""
        n = u8.shape[0]
        adt = self.net_g.engine().adt
        if not self.use_graph or n != self.B:
            return self._render_eager(state, u8)
        if self._graph is not None and self._graph[5] != adt:      # the networks switched compute dtype: the captured launches are stale
            self._graph, self._graph_src = None, None
        if self._graph is None:
            flat = (state.flat if state.flat is not None else pack_source_state(state.cache, state.src64, state.kp_source, state.kp_initial, state.scale)).clone()
            gstate = unpack_source_state(flat, adt, self.img)
            gin = torch.empty_like(self.dev_in[0])
            gin.copy_(u8)
            self._render_eager(gstate, gin)                          # warm-up outside the capture (one-time attribute calls, allocator pools)
            torch.cuda.synchronize(self.dev)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode="thread_local"):     # RCCL's watchdog thread may query events meanwhile (trainer.CAPTURE_MODE)
                gout = self._render_eager(gstate, gin)
            self._graph, self._graph_src = (g, gin, gout, flat, gstate, adt), state
        g, gin, gout, flat, gstate, _ = self._graph
        if self._graph_src is not state:
            src_flat = state.flat if state.flat is not None else pack_source_state(state.cache, state.src64, state.kp_source, state.kp_initial, state.scale)
            flat.copy_(src_flat)
            self._graph_src = state
        gin.copy_(u8)
        g.replay()
        return gout

    @torch.no_grad()
    def stream(self, state: SourceState, frames):
        """generator: yields (start_index, uint8 host tensor [n,256,256,3]) per batch, in order.  The yielded tensor is a view
        of a pinned staging buffer that is recycled two batches later: copy it if it must outlive that."""
        lazy = isinstance(frames, LazyFrames)
        if not lazy:
            frames = torch.as_tensor(frames)
        if (not lazy and (frames.dtype != torch.uint8 or frames.dim() != 4 or frames.shape[-1] != 3)) or tuple(frames.shape[1:3]) != self.hw:
            raise ValueError(f"frames must be uint8 [N,{self.hw[0]},{self.hw[1]},3], got {getattr(frames, 'dtype', 'lazy')} {tuple(frames.shape)}")
        N, B = frames.shape[0], self.B
        cur = torch.cuda.current_stream(self.dev)
        # the staging buffers came from the caching allocator, whose reuse is only ordered on the stream that allocated them:
        # work already queued on the compute stream may still be using that memory under another name, so the copy streams
        # start behind everything queued so far
        self.s_h2d.wait_stream(cur)
        self.s_d2h.wait_stream(cur)
        in_free, out_ready, pending = [None, None], [None, None], None
        nb = (N + B - 1) // B
        for i in range(nb):
            a, n, k = i * B, min(B, N - i * B), i & 1
            if in_free[k] is not None:
                in_free[k].synchronize()                         # compute has consumed the device copy staged from this host buffer
            if lazy:
                frames.copy_into(self.pin_in[k][:n], a, n)       # the decoders' output goes straight into pinned memory
            else:
                _host_copy(self.pin_in[k][:n], frames[a:a + n])  # host memcpy into pinned memory
            with torch.cuda.stream(self.s_h2d):
                self.dev_in[k][:n].copy_(self.pin_in[k][:n], non_blocking=True)
                h2d = torch.cuda.Event()
                h2d.record()
            cur.wait_event(h2d)
            out = self._render(state, self.dev_in[k][:n])
            in_free[k] = torch.cuda.Event()
            in_free[k].record(cur)
            self.dev_out[k][:n].copy_(out)                       # off the (possibly static) render output: the next batch may overwrite it
            done = torch.cuda.Event()
            done.record(cur)
            with torch.cuda.stream(self.s_d2h):                  # pin_out[k] / dev_out[k] were handed out (and consumed) one iteration ago
                self.s_d2h.wait_event(done)
                self.pin_out[k][:n].copy_(self.dev_out[k][:n], non_blocking=True)
                out_ready[k] = torch.cuda.Event()
                out_ready[k].record()
            if pending is not None:                              # hand batch i-1 to the caller while batch i runs
                pa, pn, pk = pending
                out_ready[pk].synchronize()
                yield pa, self.pin_out[pk][:pn]
            pending = (a, n, k)
        if pending is not None:
            pa, pn, pk = pending
            out_ready[pk].synchronize()
            yield pa, self.pin_out[pk][:pn]

    def run(self, state: SourceState, frames, out=None):
        """all frames -> one uint8 host tensor [N,256,256,3] (`out` may be a preallocated, e.g. pinned, destination)."""
        if not isinstance(frames, LazyFrames):
            frames = torch.as_tensor(frames)
        if out is None:
            out = torch.empty((frames.shape[0], self.img, self.img, 3), dtype=torch.uint8)
        for a, chunk in self.stream(state, frames):
            _host_copy(out[a:a + chunk.shape[0]], chunk)
        return out
