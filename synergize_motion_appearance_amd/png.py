"""Minimal PNG codec (zlib + numpy) for the host I/O either side of the animation loop: this
image has neither cv2 nor imageio, and the reference's dataset / writer path
(`basicsr/utils/img_util.py:101-155`, `basicsr/data/frames_dataset.py:244-262`) is cv2-only.
8-bit, non-interlaced, colour types 0 (gray), 2 (RGB), 3 (palette), 4 (gray+alpha), 6 (RGBA).
Arrays are in FILE order (RGB); the BGR convention of cv2 is applied by the callers in img_util."""
import os
import struct
import zlib
from concurrent.futures import ThreadPoolExecutor

import numpy as np

_SIG = b"\x89PNG\r\n\x1a\n"


def _chunk(tag: bytes, data: bytes) -> bytes:
    return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)


def default_workers() -> int:
    """codec threads: zlib and the library's scanline loop run without the interpreter lock, so the pool scales with the host's cores."""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 4
    return max(2, min(32, n))


_POOL = None


def pool() -> ThreadPoolExecutor:
    """the process-wide codec pool (created on first use)."""
    global _POOL
    if _POOL is None:
        _POOL = ThreadPoolExecutor(max_workers=default_workers(), thread_name_prefix="smx-png")
    return _POOL


def write_png(path, img: np.ndarray, level: int = 1) -> None:
    with open(path, "wb") as f:
        f.write(encode_png(img, level))


def read_png(path) -> np.ndarray:
    with open(path, "rb") as f:
        return decode_png(f.read())


def encode_many(frames, paths, level: int = 1):
    """frames[i] -> paths[i] on the codec pool; returns when every file is written."""
    futs = [pool().submit(write_png, p, np.asarray(fr), level) for fr, p in zip(frames, paths)]
    for f in futs:
        f.result()


def decode_many(paths):
    return list(pool().map(read_png, paths))


def encode_png(img: np.ndarray, level: int = 3) -> bytes:
    """uint8 [H,W] / [H,W,1|3|4] -> PNG bytes (filter 0 on every row)."""
    a = np.ascontiguousarray(img)
    if a.dtype != np.uint8:
        raise TypeError(f"PNG writer takes uint8, got {a.dtype}")
    if a.ndim == 2:
        a = a[:, :, None]
    if a.ndim != 3 or a.shape[2] not in (1, 3, 4):
        raise ValueError(f"PNG writer takes [H,W], [H,W,1], [H,W,3] or [H,W,4], got {img.shape}")
    h, w, c = a.shape
    ctype = {1: 0, 3: 2, 4: 6}[c]
    raw = np.empty((h, 1 + w * c), np.uint8)
    raw[:, 0] = 0
    raw[:, 1:] = a.reshape(h, w * c)
    ihdr = struct.pack(">IIBBBBB", w, h, 8, ctype, 0, 0, 0)
    return _SIG + _chunk(b"IHDR", ihdr) + _chunk(b"IDAT", zlib.compress(raw.tobytes(), level)) + _chunk(b"IEND", b"")


def _unfilter(raw: np.ndarray, h: int, stride: int, bpp: int) -> np.ndarray:
    """scanline reconstruction: the library's C loop (smx_png_unfilter_u8: no interpreter lock, ~0.2 ms per 256x256 RGB frame) when libsmx.so is
    there, the numpy / Python restatement below otherwise (Paeth rows cost ~1 ms per row in Python)."""
    if not raw[:, 0].any():                             # every row filter 0 (this module's own writer): a copy
        return np.ascontiguousarray(raw[:, 1:])
    try:
        from . import lib as L
        so = L.load()
    except Exception:                                   # noqa: BLE001 -- no library (a CPU-only checkout): the restatement below
        so = None
    if so is not None:
        rawc = np.ascontiguousarray(raw)
        out = np.empty((h, stride), np.uint8)
        if so.smx_png_unfilter_u8(rawc.ctypes.data, h, stride, bpp, out.ctypes.data) != 0:
            raise ValueError("bad PNG filter type")
        return out
    return _unfilter_py(raw, h, stride, bpp)


def _unfilter_py(raw: np.ndarray, h: int, stride: int, bpp: int) -> np.ndarray:
    out = np.zeros((h, stride), np.uint8)
    prev = np.zeros(stride, np.int32)
    for y in range(h):
        ft = int(raw[y, 0])
        line = raw[y, 1:].astype(np.int32)
        if ft == 0:
            cur = line
        elif ft == 2:
            cur = (line + prev) & 255
        elif ft == 1:                                   # Sub: prefix sum per byte lane
            cur = line.copy()
            for k in range(bpp):
                cur[k::bpp] = np.cumsum(line[k::bpp]) & 255
        elif ft in (3, 4):                              # Average / Paeth: sequential in x
            cur = np.zeros(stride, np.int32)
            for x in range(stride):
                a = cur[x - bpp] if x >= bpp else 0
                b = prev[x]
                if ft == 3:
                    pred = (a + b) >> 1
                else:
                    c = prev[x - bpp] if x >= bpp else 0
                    p = a + b - c
                    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
                    pred = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                cur[x] = (line[x] + pred) & 255
        else:
            raise ValueError(f"bad PNG filter type {ft}")
        out[y] = cur
        prev = cur
    return out


def decode_png(content: bytes) -> np.ndarray:
    """PNG bytes -> uint8 [H,W] (gray) or [H,W,3|4] in RGB(A) order."""
    if content[:8] != _SIG:
        raise ValueError("not a PNG stream")
    pos, idat, plte, ihdr = 8, [], None, None
    while pos < len(content):
        n, tag = struct.unpack(">I4s", content[pos:pos + 8])
        data = content[pos + 8:pos + 8 + n]
        pos += 12 + n
        if tag == b"IHDR":
            ihdr = struct.unpack(">IIBBBBB", data)
        elif tag == b"PLTE":
            plte = np.frombuffer(data, np.uint8).reshape(-1, 3)
        elif tag == b"IDAT":
            idat.append(data)
        elif tag == b"IEND":
            break
    if ihdr is None:
        raise ValueError("PNG without IHDR")
    w, h, depth, ctype, _, _, interlace = ihdr
    if depth != 8 or interlace != 0 or ctype not in (0, 2, 3, 4, 6):
        raise ValueError(f"unsupported PNG (depth {depth}, colour type {ctype}, interlace {interlace})")
    c = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ctype]
    raw = np.frombuffer(zlib.decompress(b"".join(idat)), np.uint8).reshape(h, 1 + w * c)
    px = _unfilter(raw, h, w * c, c).reshape(h, w, c)
    if ctype == 3:
        if plte is None:
            raise ValueError("palette PNG without PLTE")
        return plte[px[:, :, 0]]
    if ctype == 0:
        return px[:, :, 0]
    if ctype == 4:
        return px[:, :, 0]                              # drop alpha of gray+alpha
    return px
