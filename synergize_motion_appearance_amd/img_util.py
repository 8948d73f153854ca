"""img2tensor / tensor2img of the demo path (reference `basicsr/utils/img_util.py:13-98`),
without cv2: HWC<->CHW, optional channel flip, clamp -> [0,1] -> x255 -> round -> uint8."""
import math

import numpy as np
import torch


def img2tensor(imgs, bgr2rgb=True, float32=True):
    def one(img):
        if img.shape[2] == 3 and bgr2rgb:
            if img.dtype == "float64":
                img = img.astype("float32")
            img = img[:, :, ::-1]
        t = torch.from_numpy(np.ascontiguousarray(img.transpose(2, 0, 1)))
        return t.float() if float32 else t
    return [one(i) for i in imgs] if isinstance(imgs, list) else one(imgs)


def tensor2img(tensor, rgb2bgr=True, out_type=np.uint8, min_max=(0, 1)):
    if not (torch.is_tensor(tensor) or (isinstance(tensor, list) and all(torch.is_tensor(t) for t in tensor))):
        raise TypeError(f"tensor or list of tensors expected, got {type(tensor)}")
    tensors = [tensor] if torch.is_tensor(tensor) else tensor
    result = []
    for t in tensors:
        t = t.squeeze(0).float().detach().cpu().clamp_(*min_max)
        t = (t - min_max[0]) / (min_max[1] - min_max[0])
        if t.dim() == 4:
            n, c, h, w = t.shape
            nrow = int(math.sqrt(n))
            rows = int(math.ceil(n / nrow))
            pad = 2                                   # torchvision.utils.make_grid default padding
            grid = torch.zeros(c, rows * (h + pad) + pad, nrow * (w + pad) + pad)
            for i in range(n):
                r, cc = divmod(i, nrow)
                grid[:, pad + r * (h + pad): pad + r * (h + pad) + h, pad + cc * (w + pad): pad + cc * (w + pad) + w] = t[i]
            img = grid.numpy().transpose(1, 2, 0)
            if rgb2bgr:
                img = img[:, :, ::-1]
        elif t.dim() == 3:
            img = t.numpy().transpose(1, 2, 0)
            if img.shape[2] == 1:
                img = np.squeeze(img, axis=2)
            elif rgb2bgr:
                img = img[:, :, ::-1]
        elif t.dim() == 2:
            img = t.numpy()
        else:
            raise TypeError(f"Only support 4D, 3D or 2D tensor. But received with dimension: {t.dim()}")
        if out_type == np.uint8:
            img = (img * 255.0).round()
        result.append(np.ascontiguousarray(img.astype(out_type)))
    return result[0] if len(result) == 1 else result


# ---- host image I/O either side of the loop (reference img_util.py:101-173; cv2/imageio-free) ------
def imfrombytes(content, flag="color", float32=False):
    """PNG bytes -> BGR HWC array like cv2.imdecode (reference `imfrombytes`, img_util.py:101-127);
    float32=True scales to [0,1]. Only PNG streams are decodable without cv2."""
    from .png import decode_png
    img = decode_png(bytes(content))
    if flag == "grayscale":
        if img.ndim == 3:
            img = np.round(img[..., :3].astype(np.float32) @ np.array([0.299, 0.587, 0.114], np.float32)).astype(np.uint8)
    elif flag == "color":
        img = np.repeat(img[:, :, None], 3, 2) if img.ndim == 2 else img[..., :3]
        img = img[:, :, ::-1]                       # file order is RGB, cv2 hands out BGR
    elif flag == "unchanged":
        if img.ndim == 3:
            img = np.concatenate([img[..., 2::-1], img[..., 3:]], axis=2)
    else:
        raise ValueError(f"unknown imread flag {flag!r}")
    img = np.ascontiguousarray(img)
    return img.astype(np.float32) / 255.0 if float32 else img


def imwrite(img, file_path, params=None, auto_mkdir=True):
    """BGR (or gray) uint8 HWC array -> PNG file (reference `imwrite`, img_util.py:139-155, which
    defers to cv2.imwrite: arrays are BGR in memory, RGB in the file)."""
    import os
    from .png import encode_png
    if not str(file_path).lower().endswith(".png"):
        raise ValueError("the cv2-free writer emits PNG only")
    if auto_mkdir:
        os.makedirs(os.path.abspath(os.path.dirname(file_path)), exist_ok=True)
    a = np.asarray(img)
    if a.ndim == 3 and a.shape[2] >= 3:
        a = np.concatenate([a[..., 2::-1], a[..., 3:]], axis=2)
    with open(file_path, "wb") as f:
        f.write(encode_png(a))
    return True


def mimsave(visualizations, file_path, auto_mkdir=True, fps=None):
    """list of RGB uint8 frames -> video file (reference `mimsave`, img_util.py:157-173 =
    imageio.mimwrite; `fps` as demo.py:222 passes it). Without imageio the frames are written as
    `<file_path>.frames/%06d.png` and that folder's path is returned."""
    import os
    if auto_mkdir:
        os.makedirs(os.path.abspath(os.path.dirname(file_path)), exist_ok=True)
    try:
        import imageio
    except ImportError:
        from .png import encode_many
        d = str(file_path) + ".frames"
        os.makedirs(d, exist_ok=True)
        frames = list(visualizations)
        encode_many(frames, [os.path.join(d, f"{i:06d}.png") for i in range(len(frames))])     # on the codec thread pool (zlib releases the interpreter lock)
        return d
    return imageio.mimwrite(file_path, visualizations, **({} if fps is None else {"fps": fps}))


def resize_linear(img, size):
    """cv2.resize(img, (W,H), interpolation=INTER_LINEAR) semantics on a float HWC array:
    half-pixel centres, border clamp, no antialiasing (frames_dataset.py:264-267)."""
    w_out, h_out = int(size[0]), int(size[1])
    h, w = img.shape[:2]

    def taps(n_out, n_in):
        x = (np.arange(n_out, dtype=np.float64) + 0.5) * (n_in / n_out) - 0.5
        x0 = np.floor(x)
        f = (x - x0).astype(np.float32)
        i0 = np.clip(x0.astype(np.int64), 0, n_in - 1)
        i1 = np.clip(x0.astype(np.int64) + 1, 0, n_in - 1)
        return i0, i1, f
    y0, y1, fy = taps(h_out, h)
    x0, x1, fx = taps(w_out, w)
    a = img.astype(np.float32)
    fy = fy.reshape(-1, 1, *([1] * (a.ndim - 2)))
    fx = fx.reshape(1, -1, *([1] * (a.ndim - 2)))
    top = a[y0][:, x0] * (1 - fx) + a[y0][:, x1] * fx
    bot = a[y1][:, x0] * (1 - fx) + a[y1][:, x1] * fx
    return top * (1 - fy) + bot * fy
