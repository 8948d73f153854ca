"""A reverse-mode tape for the HIP training step (reference: `l_g_total.backward()` in
`models/appmotioncomp_model.py:418`).

The inference engines launch C-ABI kernels on raw device pointers and record nothing.  The training engines call the
differentiable wrappers of `train_ops.py`; each wrapper launches the forward kernel(s) and appends ONE closure to the tape that
launches the matching backward kernel(s) (`include/smx.h`, "TRAINING STEP").  `Tape.backward()` runs the closures in reverse.

PyTorch is the memory allocator and the stream here, nothing else: no autograd graph is built inside a network; gradients are
accumulated by this tape with the library's own kernels.  A whole network's step is exposed to torch as ONE
`torch.autograd.Function` (models.py) so that `loss.backward()` in user code reaches the parameters' `.grad`.

Bookkeeping rules:
  * gradients are keyed by tensor OBJECT (id); every op output is a fresh tensor, channel slices / concatenations are explicit
    ops (`train_ops.slice_ch`, `cat`), so two views of the same memory never alias silently;
  * `acc(t, g, owned)`: the first contribution is adopted (no copy); a second one allocates the sum unless the stored buffer is
    owned by the tape (then it is updated in place) -- a buffer that is also some OTHER tensor's gradient is never written;
  * tensors that need no gradient (inputs, detached branches, integer outputs) are registered with `stop()`; wrappers skip the
    data-gradient launch for them.
"""
import torch

from . import lib as L


class Tape:
    def __init__(self, params, grads, mfma16=False, plan=None, reduce_plan=None):
        """params / grads: {name: device tensor}; grads[name] has the parameter's shape (usually a view of one flat buffer) and is
        ACCUMULATED into by the weight-gradient kernels (zero it before the step).
        mfma16: the bf16-compute training mode (BASELINE configs[4] "bf16"): every convolution / Linear contraction -- forward, data
        gradient and weight gradient -- runs on the bf16 MFMA with operands rounded to bfloat16 on the way in and fp32 accumulation
        (torch.autocast(bfloat16)'s arithmetic for those ops); activations, gradients, parameters, normalisation, softmax / attention,
        warps and the optimiser stay fp32 (autocast additionally rounds the STORED conv outputs to bf16; this mode does not)."""
        self.P, self.G = params, grads
        self.mfma16 = bool(mfma16)
        self.nodes = []
        self._g = {}            # id(tensor) -> [grad tensor, owned by the tape?]
        self._keep = {}         # id(tensor) -> tensor (keeps ids stable while a gradient is pending)
        self._stop = set()
        self.packed = {}        # per-step cache of packed weights: (key, mode) -> tensor
        self.lib = L.load()
        self.plan = plan        # train_ops.PackPlan (optional): all packings it has seen are refreshed by ONE launch, now
        if plan is not None:
            plan.begin(self)
        self.reduce_plan = reduce_plan      # train_ops.ReducePlan (optional): the weight gradients' split reduces, one launch per backward piece
        if reduce_plan is not None:
            reduce_plan.begin(self)

    # ---- graph ----
    def record(self, fn):
        self.nodes.append(fn)

    def stop(self, t):
        """t needs no gradient (leaf input / detached branch)."""
        self._stop.add(id(t))
        self._keep[id(t)] = t
        return t

    def needs(self, t):
        return t is not None and id(t) not in self._stop

    # ---- gradients ----
    def grad(self, t):
        e = self._g.get(id(t))
        return None if e is None else e[0]

    def take(self, t):
        """gradient of t, removed from the table (its consumer runs exactly once)."""
        e = self._g.pop(id(t), None)
        self._keep.pop(id(t), None)
        return None if e is None else e[0]

    def acc(self, t, g, owned=True):
        """t.grad += g.  `owned`: the caller hands g over (nobody else reads or writes it afterwards)."""
        if t is None or g is None or id(t) in self._stop:
            return
        if tuple(g.shape) != tuple(t.shape):
            raise L.SmxError(f"tape.acc: gradient shape {tuple(g.shape)} != tensor shape {tuple(t.shape)}")
        e = self._g.get(id(t))
        if e is None:
            if not g.is_contiguous():
                g, owned = _dense(self.lib, g), True
            self._g[id(t)] = [g, owned]
            self._keep[id(t)] = t
            return
        if e[1]:
            _axpy(self.lib, g, e[0], 1.0)
        else:
            s = _dense(self.lib, e[0])
            _axpy(self.lib, g, s, 1.0)
            e[0], e[1] = s, True

    def mark(self):
        """a position in the recording: `backward(stop_at=mark)` runs exactly the nodes recorded after it."""
        return len(self.nodes)

    def backward(self, stop_at=0):
        """run the recorded closures in reverse, down to (not including) position `stop_at`; a later call continues from there."""
        while len(self.nodes) > stop_at:
            self.nodes.pop()()
        if self.reduce_plan is not None:
            self.reduce_plan.flush(self)


def _stream():
    import ctypes as C
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _pix(t):
    """(ptr, ld, P, C) of a tensor whose last dim is dense and whose leading dims are dense at a common row stride."""
    if t.stride(-1) != 1:
        raise L.SmxError("tape: innermost stride must be 1")
    C_ = t.shape[-1]
    if t.dim() == 1:
        return t.data_ptr(), C_, 1, C_
    ld = t.stride(-2) if t.shape[-2] > 1 else max(C_, t.stride(-2))
    exp = ld
    for d in range(t.dim() - 2, -1, -1):
        if t.shape[d] > 1 and t.stride(d) != exp:
            raise L.SmxError(f"tape: non-dense row layout {tuple(t.shape)} / {t.stride()}")
        exp *= t.shape[d]
    return t.data_ptr(), int(ld), t.numel() // C_, C_


def _dense(lib, t):
    out = torch.empty(t.shape, device=t.device, dtype=torch.float32)
    xp, ldx, P, C_ = _pix(t)
    L.check(lib.smx_convert_slice(xp, 0, ldx, out.data_ptr(), 0, C_, P, C_, _stream()), "copy")
    return out


def _axpy(lib, x, y, alpha):
    """y += alpha * x (row-strided views allowed)."""
    xp, ldx, P, C_ = _pix(x)
    yp, ldy, P2, C2 = _pix(y)
    if (P, C_) != (P2, C2):
        raise L.SmxError(f"tape: axpy shape mismatch {tuple(x.shape)} vs {tuple(y.shape)}")
    L.check(lib.smx_axpy_slice_f32(xp, ldx, yp, ldy, P, C_, float(alpha), _stream()), "axpy_slice")
