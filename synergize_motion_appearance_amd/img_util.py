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
