"""The generator-side training step on the HIP path (BASELINE configs[4]; reference
`models/appmotioncomp_model.py:294-434` `optimize_parameters`): forward of the training branch with a gradient tape
(`engine_train.NetGTrainEngine`), the losses that need no downloaded network, `Tape.backward()`, gradient all-reduce over
`torch.distributed` (RCCL on the GPU box: the DDP of `models/base_model.py:71-74` as bucketed all-reduces of ONE flat buffer),
a fused Adam step and the EMA copy -- all of it C-ABI kernels (`include/smx.h`, "TRAINING STEP").

`FlatParams` puts every parameter of a network into one fp32 device buffer (256-B aligned slots); the module's `nn.Parameter`s
are re-pointed at views of it, so `state_dict()` / checkpoints / `load_state_dict` keep the reference's names and shapes while
the optimiser, the EMA and the collectives see three flat arrays (values, gradients, moments): one Adam launch and a few large
all-reduces per step instead of 472 small ones (xGMI rings are per-link bound: big buckets)."""
import torch

from . import lib as L
from . import ops
from . import train_ops as T
from .engine_train import NetGTrainEngine
from .tape import Tape, _stream


class FlatParams:
    ALIGN = 64      # floats (256 B): every slot is float4-aligned for the vector kernels

    def __init__(self, module, allow_cpu=False):
        """allow_cpu: host tensors are accepted for the bookkeeping / collective logic only (the CPU `gloo` tests); `adam_step` and
        `ema_into` are HIP kernels and raise without a device."""
        named = [(n, p) for n, p in module.named_parameters()]
        if not named:
            raise ValueError("FlatParams: the module has no parameters")
        dev = named[0][1].device
        if dev.type != "cuda" and not allow_cpu:
            raise L.SmxError("FlatParams: parameters must be on the MI355X (call .cuda()); there is no CPU training path")
        offs, off = [], 0
        for _, p in named:
            offs.append(off)
            off += (p.numel() + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        self.numel = off
        self.value = torch.zeros(off, device=dev, dtype=torch.float32)
        self.grad = torch.zeros(off, device=dev, dtype=torch.float32)
        self.m = torch.zeros(off, device=dev, dtype=torch.float32)
        self.v = torch.zeros(off, device=dev, dtype=torch.float32)
        self.P, self.G, self.slots = {}, {}, {}
        for (n, p), o in zip(named, offs):
            view = self.value[o:o + p.numel()].view(p.shape)
            view.copy_(p.data)
            p.data = view                                    # the module's parameter now IS the slot
            gview = self.grad[o:o + p.numel()].view(p.shape)
            p.grad = gview
            self.P[n], self.G[n], self.slots[n] = view, gview, (o, p.numel())
        self.t = 0
        self.lib = L.load() if dev.type == "cuda" else None

    def zero_grad(self):
        self.grad.zero_()

    def all_reduce_start(self, dist, group=None, bucket_mb=64):
        """issue the sum of the flat gradient over the ranks in a few large buckets and return the pending work handles: RCCL runs them
        on its own stream behind everything already queued on the current one, so kernels launched AFTER this call (the other
        network's backward) overlap the collective.  `all_reduce_wait` joins them.  The 1/world factor is applied inside the Adam
        kernel (gscale).  gloo carrying device tensors (the one-device tests) is host-staged and completes here."""
        n = max(1, int(bucket_mb * (1 << 20) // 4))
        cpu = dist.get_backend(group) == "gloo" and self.grad.is_cuda
        works = []
        for a in range(0, self.numel, n):
            chunk = self.grad[a:a + n]
            if cpu:
                h = chunk.cpu()
                dist.all_reduce(h, group=group)
                chunk.copy_(h)
            else:
                works.append(dist.all_reduce(chunk, group=group, async_op=True))
        return works

    @staticmethod
    def all_reduce_wait(works):
        for w in works:
            w.wait()

    def all_reduce(self, dist, group=None, bucket_mb=64):
        self.all_reduce_wait(self.all_reduce_start(dist, group, bucket_mb))

    def broadcast(self, dist, src=0, group=None):
        """every rank takes `src`'s parameters, Adam moments and step counter (what DDP's constructor does for the parameters,
        models/base_model.py:71-74; the moments and the counter matter after a resume that only `src` performed)."""
        cpu = dist.get_backend(group) == "gloo" and self.value.is_cuda
        t = torch.tensor([float(self.t)], dtype=torch.float64, device="cpu" if (cpu or not self.value.is_cuda) else self.value.device)
        for buf in (self.value, self.m, self.v, t):
            if cpu and buf.is_cuda:
                h = buf.cpu()
                dist.broadcast(h, src=src, group=group)
                buf.copy_(h)
            else:
                dist.broadcast(buf, src=src, group=group)
        self.t = int(t.item())

    def adam_step(self, lr, betas=(0.9, 0.99), eps=1e-8, weight_decay=0.0, gscale=1.0):
        if self.lib is None:
            raise L.SmxError("FlatParams.adam_step is a HIP kernel: the parameters must be on the MI355X")
        self.t += 1
        L.check(self.lib.smx_adam_step_f32(self.value.data_ptr(), self.grad.data_ptr(), self.m.data_ptr(), self.v.data_ptr(), self.numel,
                                           float(lr), float(betas[0]), float(betas[1]), float(eps), float(weight_decay), self.t, float(gscale),
                                           _stream()), "adam_step")

    def optimizer_state_dict(self, lr, betas=(0.9, 0.99), eps=1e-8, weight_decay=0.0):
        """this buffer's Adam state in torch.optim.Adam's own state_dict layout (per parameter, in named_parameters order -- the order
        the reference builds its optimizers in, models/appmotioncomp_model.py:235-270): a `.state` file written from it resumes a
        torch.optim.Adam, and one written by the reference resumes this."""
        state = {}
        for i, (name, (off, n)) in enumerate(self.slots.items()):
            shape = self.P[name].shape
            state[i] = {"step": torch.tensor(float(self.t)), "exp_avg": self.m[off:off + n].view(shape).detach().cpu().clone(),
                        "exp_avg_sq": self.v[off:off + n].view(shape).detach().cpu().clone()}
        group = {"lr": float(lr), "betas": tuple(betas), "eps": float(eps), "weight_decay": float(weight_decay), "amsgrad": False,
                 "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
                 "params": list(range(len(self.slots)))}
        return {"state": state if self.t > 0 else {}, "param_groups": [group]}

    def load_optimizer_state_dict(self, sd):
        """-> the learning rate stored in the file (the caller's schedule decides whether to use it)."""
        st = sd.get("state", {})
        names = list(self.slots)
        if st and (len(st) > len(names) or any(int(i) >= len(names) for i in st)):
            raise ValueError(f"optimizer state has {len(st)} parameters, this network has {len(names)}")
        # torch.optim.Adam omits the entry of a parameter that never received a gradient: a missing index = zero moments
        self.m.zero_()
        self.v.zero_()
        self.t = 0
        for i, name in enumerate(names):
            if i not in st:
                continue
            off, n = self.slots[name]
            e = st[i]
            if tuple(e["exp_avg"].shape) != tuple(self.P[name].shape):
                raise ValueError(f"optimizer state of parameter {i} ({name}) has shape {tuple(e['exp_avg'].shape)}, expected {tuple(self.P[name].shape)}")
            self.m[off:off + n].copy_(e["exp_avg"].reshape(-1).to(self.m.device, torch.float32))
            self.v[off:off + n].copy_(e["exp_avg_sq"].reshape(-1).to(self.v.device, torch.float32))
            self.t = int(float(e["step"]))
        return float(sd["param_groups"][0]["lr"]) if sd.get("param_groups") else None

    def ema_into(self, other, decay):
        """other.value = decay * other.value + (1 - decay) * self.value (models/sr_model.py model_ema); same slot layout required."""
        if other.numel != self.numel:
            raise ValueError("EMA copy has a different parameter layout")
        L.check(self.lib.smx_ema_f32(other.value.data_ptr(), self.value.data_ptr(), self.numel, float(decay), _stream()), "ema")


class NetGTrainStep:
    """One generator step: losses of `optimize_parameters` (:308-386) that need no downloaded network.
    train_opt: the yml's `train` dict; used keys: pixel_opt, motion_codebook_code_opt, motion_codebook_recon_opt,
    lr_pixel_perceptual_opt, app_codebook_code_opt, optim_g, ema_decay."""

    def __init__(self, net_g, train_opt, compute_dtype="f32"):
        """compute_dtype "bf16": every convolution / Linear contraction of the step (forward, data and weight gradient) on the bf16
        MFMA, everything else fp32 (Tape(mfma16=True)); yml: `train.compute_dtype: bf16`."""
        self.net_g = net_g
        self.opt = dict(train_opt)
        if compute_dtype not in ("f32", "bf16"):
            raise ValueError(f"compute_dtype {compute_dtype!r}: f32 or bf16")
        self.mfma16 = compute_dtype == "bf16"
        self._pack_plan = T.PackPlan()          # this step's weight packings, refreshed by one launch per step from the second step on
        self.flat = FlatParams(net_g)
        net_g.refresh()                                      # the inference engine's packed copies are stale from now on
        self.engine = NetGTrainEngine(net_g.cfg)
        og = dict(self.opt.get("optim_g", {}))
        og.pop("type", None)
        self.lr, self.betas = float(og.get("lr", 8e-5)), tuple(og.get("betas", (0.9, 0.99)))
        self.wd, self.eps = float(og.get("weight_decay", 0)), float(og.get("eps", 1e-8))

    @staticmethod
    def _w(opt, key, default=0.0):
        o = opt.get(key)
        if not o:
            return default
        lw = o.get("loss_weight", 1.0)
        return lw

    def forward_backward(self, source, driving, dense_motion, w=1.0, backward=True):
        """source / driving [B,3,256,256] in [-1,1]; dense_motion: deformation [B,64,64,2], occlusion_map [B,1,64,64],
        driving_kp_heatmap [B,15,64,64] (device tensors).  -> (loss_dict of device scalars incl. 'l_g_total', out_dict with NCHW
        'out', gradients w.r.t. the three dense-motion inputs in their own layouts).  Parameter gradients accumulate into
        `self.flat.grad` (call `flat.zero_grad()` first)."""
        flat = self.flat
        tp = Tape(flat.P, flat.G, mfma16=self.mfma16, plan=self._pack_plan)
        B = driving.shape[0]
        defo = dense_motion["deformation"].float().contiguous()
        occ = dense_motion["occlusion_map"].float().reshape(B, 64, 64).contiguous()
        heat = ops.nchw_to_nhwc(dense_motion["driving_kp_heatmap"].float())
        st = self.engine.forward(tp, source.float(), defo, occ, heat, float(w), gt_nchw=driving.float())
        gt = tp.stop(ops.nchw_to_nhwc(driving.float()))
        o = self.opt
        terms, losses = [], {}

        def add(name, t, weight=1.0):
            losses[name] = t
            terms.append((t, weight))
        if o.get("pixel_opt"):
            add("l_g_pix", T.l1_loss(tp, st["out"], gt, self._w(o, "pixel_opt", 1.0)))
        wc = self._w(o, "motion_codebook_code_opt", 1.0)
        if wc:
            add("l_g_motion_codebook_code", T.weighted_sum(tp, [(l, wc) for l in st["train"]["loss_motion"]]))
        if o.get("motion_codebook_recon_opt"):
            wr = self._w(o, "motion_codebook_recon_opt", 1.0)
            recs = []
            for i, rec in enumerate(st["train"]["motion_recon"]):      # L1(m_recon / 31.5, (deformation_list[i] - grid).detach()) (:342-352)
                tgt = tp.stop(T.scaled(tp, ops.flow_to_residual(st["flows"][i]), 1.0 / 31.5))
                recs.append((T.l1_loss(tp, T.scale(tp, rec, 1.0 / 31.5), tgt, wr), 1.0))
            add("l_g_motion_codebook_recon", T.weighted_sum(tp, recs))
        lrw = (o.get("lr_pixel_perceptual_opt") or {}).get("loss_weight", [])
        if len(lrw) > 0 and o.get("pixel_opt"):
            add("l_g_pix_lr_0", T.l1_loss(tp, st["out_lr"], gt, self._w(o, "pixel_opt", 1.0) * float(lrw[0])))
        wa = self._w(o, "app_codebook_code_opt", 1.0)
        if wa > 0:
            add("l_g_app_codebook_code", T.weighted_sum(tp, [(l, wa) for l in st["app"][1]]))
        total = T.weighted_sum(tp, terms)
        losses["l_g_total"] = total
        grads_in = None
        if backward:
            tp.acc(total, torch.ones(1, device=total.device))
            tp.backward()
            gd, go, gh = tp.take(defo), tp.take(occ), tp.take(heat)
            grads_in = {"deformation": gd, "occlusion_map": None if go is None else go.view(B, 1, 64, 64),
                        "driving_kp_heatmap": None if gh is None else ops.nhwc_to_nchw(gh)}
        out = {"out": ops.nhwc_to_nchw(st["out"]), "out_lr": [ops.nhwc_to_nchw(st["out_lr"])], "deformation_list": st["flows"],
               "out_occ": [t.view(B, 1, 64, 64) for t in st["occ"][1:]],
               "codebook_loss_motion_list": st["train"]["loss_motion"], "codebook_loss_app_list": st["app"][1],
               "_vq_stats_motion": st["train"]["stats_motion"], "_vq_stats_app": st["app"][2]}
        return losses, out, grads_in

    def step(self, source, driving, dense_motion, w=1.0, ema=None, ema_decay=0.0):
        """zero_grad -> forward/backward -> (all-reduce) -> Adam -> (EMA).  -> (loss_dict, out_dict, input gradients)"""
        import torch.distributed as dist
        self.flat.zero_grad()
        losses, out, grads_in = self.forward_backward(source, driving, dense_motion, w)
        world = 1
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            world = dist.get_world_size()
            self.flat.all_reduce(dist)
        self.flat.adam_step(self.lr, self.betas, self.eps, self.wd, gscale=1.0 / world)
        if ema is not None and ema_decay > 0:
            self.flat.ema_into(ema, ema_decay)
        return losses, out, grads_in


class EquivarianceTransform:
    """The random affine + thin-plate warp of the equivariance constraint (reference `Transform`,
    models/appmotioncomp_model.py:50-104): theta = I + N(0, sigma_affine) [B,2,3], a points x points control grid on [-1,1]^2 with weights
    N(0, sigma_tps) [B,1,points^2].  The FRAME is warped by a HIP kernel (`smx_tps_transform_frame_f32`); the 15 keypoints per sample go
    through the same map with a handful of torch ops (30 numbers per sample: host-side bookkeeping like the hull scale, not a hot path)."""

    def __init__(self, bs, sigma_affine=0.05, sigma_tps=None, points_tps=None, theta=None, control_params=None, device="cuda", generator=None):
        dev = torch.device(device)
        if theta is None:
            theta = torch.normal(mean=0, std=sigma_affine * torch.ones([bs, 2, 3]), generator=generator) + torch.eye(2, 3).view(1, 2, 3)
        self.theta = theta.to(dev).float().contiguous()
        self.bs = bs
        self.tps = (sigma_tps is not None and points_tps is not None) or control_params is not None
        if self.tps:
            n = int(points_tps) if points_tps is not None else int(round(control_params.shape[-1] ** 0.5))
            lin = 2.0 * (torch.arange(n, dtype=torch.float32) / (n - 1)) - 1.0
            yy, xx = lin.view(-1, 1).repeat(1, n), lin.view(1, -1).repeat(n, 1)
            self.control_points = torch.stack([xx, yy], -1).view(1, n * n, 2).to(dev)          # make_coordinate_grid order: (x, y), x fastest
            if control_params is None:
                control_params = torch.normal(mean=0, std=sigma_tps * torch.ones([bs, 1, n * n]), generator=generator)
            self.control_params = control_params.to(dev).float().contiguous()

    def transform_frame(self, frame_nchw):
        x = frame_nchw.float().contiguous()
        y = torch.empty_like(x)
        B, C_, H, W = x.shape
        lib = L.load()
        L.check(lib.smx_tps_transform_frame_f32(x.data_ptr(), y.data_ptr(), self.theta.data_ptr(),
                                                self.control_points.data_ptr() if self.tps else None,
                                                self.control_params.data_ptr() if self.tps else None,
                                                self.control_points.shape[1] if self.tps else 0, B, C_, H, W, _stream()), "tps_transform_frame")
        return y

    def warp_coordinates(self, coordinates):
        """coordinates [B,N,2] -> [B,N,2] (the keypoint-sized torch path)."""
        theta = self.theta.unsqueeze(1)
        out = torch.matmul(theta[:, :, :, :2], coordinates.unsqueeze(-1)).squeeze(-1) + theta[:, :, :, 2]
        if self.tps:
            d = (coordinates.view(coordinates.shape[0], -1, 1, 2) - self.control_points.view(1, 1, -1, 2)).abs().sum(-1)
            r = (d ** 2 * torch.log(d + 1e-6) * self.control_params).sum(dim=2).view(self.bs, coordinates.shape[1], 1)
            out = out + r
        return out

    def jacobian(self, coordinates):
        new = self.warp_coordinates(coordinates)
        gx = torch.autograd.grad(new[..., 0].sum(), coordinates, create_graph=True)[0]
        gy = torch.autograd.grad(new[..., 1].sum(), coordinates, create_graph=True)[0]
        return torch.cat([gx.unsqueeze(-2), gy.unsqueeze(-2)], dim=-2)


def equivariance_losses(kp_driving, kp_transformed, transform, w_value=1.0, w_jacobian=1.0):
    """EquivarianceLoss.forward (losses/losses.py:540-560) on keypoint-sized torch tensors (requires_grad leaves)."""
    lv = (kp_driving["value"] - transform.warp_coordinates(kp_transformed["value"])).abs().mean() * w_value
    jt = torch.matmul(transform.jacobian(kp_transformed["value"]), kp_transformed["jacobian"])
    # 2x2 inverse in closed form (adjugate / determinant): torch.inverse checks its `info` on the host, which a hipGraph capture of the
    # step cannot contain; same value to fp32 rounding
    J = kp_driving["jacobian"]
    a, b, c, d = J[..., 0, 0], J[..., 0, 1], J[..., 1, 0], J[..., 1, 1]
    inv = torch.stack([torch.stack([d, -b], -1), torch.stack([-c, a], -1)], -2) / (a * d - b * c).unsqueeze(-1).unsqueeze(-1)
    val = torch.matmul(inv, jt)
    eye = torch.eye(2, device=val.device).view(1, 1, 2, 2)
    lj = (eye - val).abs().mean() * w_jacobian
    return lv, lj


def kp_distance_value(kp_driving_value, kp_source_value, weight=1.0):
    """KPDistanceLoss (losses/losses.py:609-616): a torch.sign of distances -- a logged VALUE with zero gradient."""
    def one(v):
        d = v.unsqueeze(2) - v.unsqueeze(1)
        K = v.shape[1]
        return (-torch.sign((torch.sqrt((d * d).sum(-1) + 1e-8) + torch.eye(K, device=v.device) * 0.2) - 0.2) + 1).mean()
    return (one(kp_source_value) + one(kp_driving_value)) * weight


# hipGraph captures run in THREAD-LOCAL error mode: with a process group alive, RCCL's watchdog thread polls the events of earlier collectives
# (hipEventQuery) while this thread captures; under the default global mode that call is "not permitted while a stream is capturing", the
# watchdog raises and takes the process down (found by tests/test_gpu_rccl.py on one GPU -- every multi-GPU training run would have hit it).
CAPTURE_MODE = "thread_local"


class TrainStep:
    """The generator + motion-estimator half of `optimize_parameters` (models/appmotioncomp_model.py:294-420) on the HIP path:
    motion_estimator(gt, source) in training mode -> net_g(source, dense_motion, w=1, gt=gt) -> losses -> ONE backward through both
    networks -> Adam on each (optim_g / optim_motion) -> EMA of net_g.  One tape spans both networks (their parameter names are disjoint)."""

    def __init__(self, net_g, motion_estimator, train_opt, compute_dtype=None, use_graph=None, net_d=None):
        """use_graph (yml `train.use_hip_graph`): capture zero_grad + forward + losses + tape backward of one step in a hipGraph after
        `GRAPH_WARMUP` eager steps and replay it from then on (inputs and the equivariance transform's random parameters are copied
        into static buffers; all-reduce, Adam and EMA stay outside: they carry the step counter).  The step is launch-bound at the
        reference's batch sizes (~2,400 launches behind Python + ctypes); the replay removes the host from the critical path."""
        from .engine_motion_train import MotionTrainEngine
        compute_dtype = compute_dtype or str(dict(train_opt).get("compute_dtype", "f32"))
        self.use_graph = bool(dict(train_opt).get("use_hip_graph", False)) if use_graph is None else bool(use_graph)
        self._graph, self._graph2, self._static, self._eager_steps = None, None, None, 0
        self._variant = None                     # (gan, overlap cut) of the last step: a change drops the captured graphs (step())
        self.g = NetGTrainStep(net_g, train_opt, compute_dtype)
        self.me = motion_estimator
        self.flat_m = FlatParams(motion_estimator)
        motion_estimator.refresh()
        common, dense, kp = motion_estimator._cfg
        self.bufs = dict(motion_estimator.named_buffers())
        self.me_engine = MotionTrainEngine(common, dense, kp, self.bufs)
        self.opt = dict(train_opt)
        om = dict(self.opt.get("optim_motion") or {})
        om.pop("type", None)
        self.lr_m, self.betas_m = float(om.get("lr", 8e-5)), tuple(om.get("betas", (0.9, 0.99)))
        self.wd_m = float(om.get("weight_decay", 0))
        self.P = {**self.g.flat.P, **self.flat_m.P}
        self._pack_plan, self._pack_plan_d = T.PackPlan(), T.PackPlan()     # generator-side tape (incl. its pass through net_d) / discriminator step
        # the weight gradients' split reduces, one launch per backward piece from the second step on (train.batched_wgrad_reduce, default on);
        # one plan per call sequence: (GAN branch?, backward cut in two for the overlapped all-reduce?) and the discriminator's own step
        self._reduce_plans = {} if self.g.opt.get("batched_wgrad_reduce", True) else None
        self.G = {**self.g.flat.G, **self.flat_m.G}
        # the discriminator side of optimize_parameters (models/appmotioncomp_model.py:324-345, 408-432): hinge GAN with the adaptive weight,
        # active in `step(..., gan=True)` (the model turns it on past net_d_start_iter)
        self.net_d, self.flat_d = net_d, None
        if net_d is not None:
            self.flat_d = FlatParams(net_d)
            net_d.refresh()
            self.d_plan, self.d_bufs = net_d._plan, dict(net_d.named_buffers())
            od = dict(self.opt.get("optim_d") or {})
            od.pop("type", None)
            self.lr_d, self.betas_d, self.wd_d = float(od.get("lr", 8e-5)), tuple(od.get("betas", (0.9, 0.99))), float(od.get("weight_decay", 0))
            self.P_d = {"net_d." + n: v for n, v in self.flat_d.P.items()}
            self.G_d = {"net_d." + n: v for n, v in self.flat_d.G.items()}
            self.P.update(self.P_d)
            self.G.update(self.G_d)                           # written by the generator step's pass through net_d, zeroed before the discriminator step
            go = dict(self.opt.get("gan_opt") or {})
            if go.get("gan_type", "hinge") != "hinge":
                raise NotImplementedError(f"gan_opt.gan_type {go.get('gan_type')}: the shipped train.yml uses hinge; only that has a HIP plan")
            if float(go.get("loss_weight", 1.0)) != 1.0:
                raise NotImplementedError(f"gan_opt.loss_weight {go.get('loss_weight')}: the shipped train.yml uses 1.0 (the adaptive weight scales the term); "
                                          "only that has a HIP plan")
            # models/appmotioncomp_model.py:209 reads `fix_generator` with default TRUE (adaptive weight from fuse_convs_dict[largest].shift[-1].weight,
            # :337-340); the shipped train.yml sets false (generator.blocks[-1].weight, :333-335) -- the only form built here
            if self.opt.get("fix_generator", True):
                raise NotImplementedError("train.fix_generator true (the reference's default when the key is absent): the adaptive GAN weight would be "
                                          "taken at fuse_convs_dict[...].shift[-1].weight; only `fix_generator: false` (the shipped train.yml) has a HIP plan")
            self.gan_scale = float(self.opt.get("scale_adaptive_gan_weight", 0.8))
        eq = dict(self.opt.get("equivariance_opt") or {})
        if eq and not (eq.get("use_value", True) and eq.get("use_jacobian", True)):
            raise NotImplementedError("equivariance_opt.use_value / use_jacobian false: the shipped train.yml uses both terms (losses/losses.py:540-560); "
                                      "only that has a HIP plan")
        self.percep = self._build_perceptual(self.opt.get("perceptual_opt"), next(net_g.parameters()).device)
        if self.percep is not None:
            self.P.update(self.percep.P)                      # frozen: values only, no gradient slots
        # overlap_allreduce: net_g's flat gradient is complete when the tape's backward has passed the start of net_g's forward; its
        # all-reduce is issued there and runs under the estimator's backward (DDP's overlap, models/base_model.py:71-74, at network
        # granularity).  Under the hipGraph the step is captured as TWO graphs cut at that point.
        self.overlap_allreduce = bool(self.opt.get("overlap_allreduce", True))
        self._pending = []
        self.sync_replicas()

    COLLECTIVES_AT_WORLD_1 = False     # tests/test_gpu_rccl.py: run every collective call site over a 1-rank RCCL group (identities, real communicator)

    @classmethod
    def _dist(cls):
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or cls.COLLECTIVES_AT_WORLD_1):
            return dist
        return None

    def sync_replicas(self, src=0):
        """rank `src`'s parameters, Adam state and BatchNorm buffers on every rank: what DDP does at construction (parameters) and in every
        forward (buffers: `broadcast_buffers`), models/base_model.py:71-74.  Called by the constructor and after a resume; from then on the
        replicas stay bit-identical because every rank applies the same all-reduced gradient (BatchNorm running statistics are
        per-rank batch statistics exactly as under DDP without SyncBN -- only rank 0's are saved)."""
        dist = self._dist()
        if dist is None:
            return False
        flats = [self.g.flat, self.flat_m] + ([self.flat_d] if self.flat_d is not None else [])
        for f in flats:
            f.broadcast(dist, src)
        bufs = list(self.bufs.values()) + (list(self.d_bufs.values()) if self.flat_d is not None else [])
        host = dist.get_backend() == "gloo"
        for b in bufs:
            if host and b.is_cuda:
                h = b.cpu()
                dist.broadcast(h, src=src)
                b.copy_(h)
            else:
                dist.broadcast(b, src=src)
        self.g.net_g.refresh()
        self.me.refresh()
        return True

    @staticmethod
    def _build_perceptual(po, device):
        """`train.perceptual_opt` (MultiScalePyramidPerceptualLoss, models/appmotioncomp_model.py:170, 319-322, 374-377).  The VGG19 ImageNet
        weights are a download the reference does at construction (archs/vgg_arch.py:173): here they come from `perceptual_opt.vgg19_path`
        (a torch-saved state dict, torchvision `features.N.*` or reference `sliceK.N.*` keys), or -- benchmarks and tests only --
        `perceptual_opt.synthetic_vgg19: true` (name-keyed synthetic weights).  Neither: a loud error, never a silent skip."""
        if not po:
            return None
        from .perceptual import PerceptualLoss, synthetic_vgg19_state
        po = dict(po)
        if po.get("type", "MultiScalePyramidPerceptualLoss") != "MultiScalePyramidPerceptualLoss":
            raise NotImplementedError(f"perceptual_opt.type {po.get('type')}: only MultiScalePyramidPerceptualLoss has a HIP plan")
        if po.get("vgg19_path"):
            state = torch.load(po["vgg19_path"], map_location="cpu")
            state = state.get("state_dict", state) if isinstance(state, dict) else state
        elif po.get("synthetic_vgg19"):
            state = synthetic_vgg19_state()
        else:
            raise RuntimeError("train.perceptual_opt needs the torchvision VGG19 weights (the reference downloads them): set "
                               "perceptual_opt.vgg19_path to a saved vgg19 state dict, or perceptual_opt.synthetic_vgg19: true for a benchmark run")
        return PerceptualLoss(state, scales=po.get("scales", [1, 0.5, 0.25, 0.125]), loss_weights=po.get("loss_weights", [1.0] * 5), device=device)

    def _disc(self, tp, x):
        """VQGANDiscriminator.forward in train mode (archs/vqgan_arch.py:536-575): conv4x4 / LeakyReLU(0.2) / BatchNorm on BATCH statistics
        (running buffers updated like nn.BatchNorm2d, num_batches_tracked counted).  x NHWC -> logit map [B,h,w,1]."""
        from .lib import ACT_LRELU02, ACT_NONE
        h = x
        for k, (i, _, _, stride, has_bias, has_bn) in enumerate(self.d_plan):
            oh, ow = (h.shape[1] + 2 - 4) // stride + 1, (h.shape[2] + 2 - 4) // stride + 1
            fused = ACT_LRELU02 if (not has_bn and k != len(self.d_plan) - 1) else ACT_NONE
            h = T.conv(tp, h, f"net_d.main.{i}.weight", f"net_d.main.{i}.bias" if has_bias else None, stride=stride, pad=(1, 1), out_hw=(oh, ow), act=fused)
            if has_bn:
                bn = f"main.{i + 1}"
                h = T.bn_relu(tp, h, f"net_d.{bn}.weight", f"net_d.{bn}.bias", self.d_bufs[bn + ".running_mean"], self.d_bufs[bn + ".running_var"], relu=False)
                self.d_bufs[bn + ".num_batches_tracked"].add_(1)
                h = T.act(tp, h, ACT_LRELU02)
        return h

    def forward_backward(self, source, driving, w=1.0, transform=None, gan=False, on_cut=None):
        """on_cut: called once in the middle of the backward, when every gradient of net_g (and of the discriminator pass) is final and
        only the motion estimator's backward remains -- the place the generator's gradient all-reduce is issued from (and where the
        hipGraph capture of the step is cut in two)."""
        g = self.g
        if gan and self.flat_d is None:
            raise RuntimeError("TrainStep(gan=True) needs net_d (the discriminator network) at construction")
        rp = None if self._reduce_plans is None else self._reduce_plans.setdefault((bool(gan), on_cut is not None), T.ReducePlan())
        tp = Tape(self.P, self.G, mfma16=self.g.mfma16, plan=self._pack_plan, reduce_plan=rp)
        src, drv = source.float().contiguous(), driving.float().contiguous()
        B = drv.shape[0]
        eng = self.me_engine
        kp_d = eng.kp_detector(tp, drv)
        kp_s = eng.kp_detector(tp, src)
        deform, occ, heat, aux = eng.dense_motion(tp, src, kp_d, kp_s)
        cut = tp.mark()                                       # backward position: everything of net_g (and the losses) is behind it
        st = g.engine.forward(tp, src, deform, occ, heat, float(w), gt_nchw=drv)
        gt = tp.stop(ops.nchw_to_nhwc(drv))
        o = self.opt
        terms, losses = [], {}

        def add(name, t, weight=1.0):
            losses[name] = t
            terms.append((t, weight))
        out_r, gan_state = st["out"], {}
        if gan:
            # adaptive GAN weight (:222-228, :334-343): d_weight = clamp(|d recon / dW| / (|d gan / dW| + 1e-4), 0, 1) * 0.8 with W the weight of the
            # generator's last convolution and recon = perceptual + pixel loss.  Both reach W only through `out`, so the reconstruction losses
            # read one alias of `out`, the discriminator another, and the node below -- recorded BEFORE their consumers, hence run AFTER them
            # in the backward -- holds the two gradients apart, takes the two weight gradients of the last layer (one TN GEMM each),
            # forms d_weight on the device and hands `out` d recon + d_weight * d gan
            out_full = st["out"]
            out_r, out_g = out_full.view(out_full.shape), out_full.view(out_full.shape)
            last_in = st["out_last_in"]

            def split():
                gr, gg = tp.take(out_r), tp.take(out_g)
                Bq, Hq, Wq, Cq = out_full.shape
                norms = []
                for gpart in (gr, gg):
                    wgt = torch.empty((Cq, last_in.shape[-1], 3, 3), device=out_full.device, dtype=torch.float32)
                    gd = gpart if gpart.is_contiguous() else gpart.contiguous()
                    T._wgrad(tp, gd, last_in, wgt, M=Bq * Hq * Wq, cout=Cq, Hin=Hq, Win=Wq, cin=last_in.shape[-1], Ho=Hq, Wo=Wq, kh=3, kw=3,
                             stride=1, pt=1, pl=1, layout=0, dy_ld=Cq, accumulate=False)
                    norms.append(wgt.norm())
                dw = (norms[0] / (norms[1] + 1e-4)).clamp(0.0, 1.0) * self.gan_scale
                gan_state["d_weight"] = dw.reshape(1)
                gan_state["norms"] = (norms[0].reshape(1), norms[1].reshape(1))
                tp.acc(out_full, gr + dw * gg)
            tp.record(split)
        if o.get("pixel_opt"):
            add("l_g_pix", T.l1_loss(tp, out_r, gt, g._w(o, "pixel_opt", 1.0)))
        wc = g._w(o, "motion_codebook_code_opt", 1.0)
        if wc:
            add("l_g_motion_codebook_code", T.weighted_sum(tp, [(l, wc) for l in st["train"]["loss_motion"]]))
        if o.get("motion_codebook_recon_opt"):
            wr = g._w(o, "motion_codebook_recon_opt", 1.0)
            recs = []
            for i, rec in enumerate(st["train"]["motion_recon"]):
                tgt = tp.stop(T.scaled(tp, ops.flow_to_residual(st["flows"][i]), 1.0 / 31.5))
                recs.append((T.l1_loss(tp, T.scale(tp, rec, 1.0 / 31.5), tgt, wr), 1.0))
            add("l_g_motion_codebook_recon", T.weighted_sum(tp, recs))
        if self.percep is not None:
            add("l_g_percep", self.percep(tp, out_r, gt))
        if gan:
            fake_pred = self._disc(tp, out_g)
            l_gan = T.torch_scalar(tp, fake_pred, lambda p: -p.mean())     # GANLoss hinge, generator side (losses/losses.py:446-449)
            terms.append((l_gan, 1.0))                                        # its gradient is scaled by d_weight in `split`
        lrw = (o.get("lr_pixel_perceptual_opt") or {}).get("loss_weight", [])
        if len(lrw) > 0 and o.get("pixel_opt"):
            add("l_g_pix_lr_0", T.l1_loss(tp, st["out_lr"], gt, g._w(o, "pixel_opt", 1.0) * float(lrw[0])))
        if len(lrw) > 0 and self.percep is not None:
            add("l_g_percep_lr_0", self.percep(tp, st["out_lr"], gt, float(lrw[0])))
        wa = g._w(o, "app_codebook_code_opt", 1.0)
        if wa > 0:
            add("l_g_app_codebook_code", T.weighted_sum(tp, [(l, wa) for l in st["app"][1]]))
        total = T.weighted_sum(tp, terms)
        # equivariance (:388-399): third keypoint pass on the TPS-warped driving frames; the keypoint-sized loss algebra runs on torch
        # leaves whose gradients are handed to the tape
        eq = o.get("equivariance_opt")
        total_val = total
        if eq:
            if transform is None:
                tparams = dict(eq.get("transform_params", {}))
                transform = EquivarianceTransform(B, **tparams, device=drv.device)
            kp_t = eng.kp_detector(tp, transform.transform_frame(drv))
            leaves = {k: v.detach().clone().requires_grad_() for k, v in (("dv", kp_d[0]), ("dj", kp_d[1]), ("tv", kp_t[0]), ("tj", kp_t[1]))}
            with torch.enable_grad():
                lv, lj = equivariance_losses({"value": leaves["dv"], "jacobian": leaves["dj"]}, {"value": leaves["tv"], "jacobian": leaves["tj"]},
                                             transform, eq.get("loss_weight_value", 1.0), eq.get("loss_weight_jacobian", 1.0))
                (lv + lj).backward()
            losses["l_equivariance_value"], losses["l_equivariance_jacobian"] = lv.detach().view(1), lj.detach().view(1)
            for t, k in ((kp_d[0], "dv"), (kp_d[1], "dj"), (kp_t[0], "tv"), (kp_t[1], "tj")):
                tp.acc(t, leaves[k].grad.contiguous())
            total_val = total + lv.detach() + lj.detach()
        if o.get("kp_distance_opt"):
            losses["l_kpd"] = kp_distance_value(kp_d[0], kp_s[0], o["kp_distance_opt"].get("loss_weight", 1.0)).view(1)
            total_val = total_val + losses["l_kpd"]
        tp.acc(total, torch.ones(1, device=total.device))
        if on_cut is not None:
            tp.backward(stop_at=cut)
            on_cut()
        tp.backward()
        if gan:                                                # d_weight exists once the backward has passed `out`; the weighted sum above counted l_gan once
            dw = gan_state["d_weight"]
            losses["d_weight"], losses["l_g_gan"] = dw, dw * l_gan
            losses["_recon_grad_norm"], losses["_gan_grad_norm"] = gan_state["norms"]     # the two last-layer gradient norms behind d_weight
            total_val = total_val + (dw - 1.0) * l_gan
        losses["l_g_total"] = total_val
        out = {"out": ops.nhwc_to_nchw(st["out"]), "out_lr": [ops.nhwc_to_nchw(st["out_lr"])], "deformation_list": st["flows"],
               "kp_driving": {"value": kp_d[0], "jacobian": kp_d[1]}, "kp_source": {"value": kp_s[0], "jacobian": kp_s[1]},
               "deformation": deform, "occlusion_map": occ.view(B, 1, 64, 64), "driving_kp_heatmap_nhwc": heat,
               "_out_nhwc": st["out"], "_gt_nhwc": gt}
        if eq:
            out["kp_transformed"] = {"value": kp_t[0], "jacobian": kp_t[1]}
        if rp is not None:
            rp.finish()
        return losses, out

    def disc_backward(self, out_nhwc, gt_nhwc):
        """the discriminator half (:408-430): hinge losses of net_d on the real frames and on the DETACHED generated ones (two passes: BatchNorm
        sees each batch on its own), gradients accumulated into net_d's flat buffer (zero it first).  -> loss dict"""
        rp = None if self._reduce_plans is None else self._reduce_plans.setdefault("d", T.ReducePlan())
        tp = Tape(self.P_d, self.G_d, mfma16=self.g.mfma16, plan=self._pack_plan_d, reduce_plan=rp)
        real, fake = tp.stop(gt_nhwc), tp.stop(out_nhwc.detach())
        pr = self._disc(tp, real)
        l_real = T.torch_scalar(tp, pr, lambda p: torch.relu(1.0 - p).mean())
        pf = self._disc(tp, fake)
        l_fake = T.torch_scalar(tp, pf, lambda p: torch.relu(1.0 + p).mean())
        one = torch.ones(1, device=out_nhwc.device)
        tp.acc(l_real, one)
        tp.acc(l_fake, one.clone())
        tp.backward()
        if rp is not None:
            rp.finish()
        return {"l_d_real": l_real, "out_d_real": pr.mean().reshape(1), "l_d_fake": l_fake, "out_d_fake": pf.mean().reshape(1)}

    GRAPH_WARMUP = 2

    def optimizer_state_dicts(self):
        """[optimizer_g, optimizer_m (, optimizer_d)] -- the reference's `self.optimizers` order (appmotioncomp_model.py:248, 265, 270)."""
        out = [self.g.flat.optimizer_state_dict(self.g.lr, self.g.betas, self.g.eps, self.g.wd),
               self.flat_m.optimizer_state_dict(self.lr_m, self.betas_m, 1e-8, self.wd_m)]
        if self.flat_d is not None:
            out.append(self.flat_d.optimizer_state_dict(self.lr_d, self.betas_d, 1e-8, self.wd_d))
        return out

    def load_optimizer_state_dicts(self, sds):
        flats = [self.g.flat, self.flat_m] + ([self.flat_d] if self.flat_d is not None else [])
        if len(sds) != len(flats):
            raise ValueError(f"Wrong lengths of optimizers: the state file has {len(sds)}, this step has {len(flats)}")
        for f, sd in zip(flats, sds):
            f.load_optimizer_state_dict(sd)

    def _draw_transform(self, B, dev):
        eq = self.opt.get("equivariance_opt")
        return EquivarianceTransform(B, **dict(eq.get("transform_params", {})), device=dev) if eq else None

    def _graph_step(self, source, driving, w, transform, gan=False, on_cut=None):
        """replay (capturing first) the hipGraph of zero_grad + forward_backward for this input shape.  With `on_cut` the step is TWO
        graphs sharing one memory pool -- [zero_grad, forwards, losses, backward of net_g] and [backward of the motion estimator] --
        and `on_cut()` (the generator's gradient all-reduce) is issued between their replays."""
        st = self._static
        key = (tuple(source.shape), tuple(driving.shape), float(w), bool(gan), on_cut is not None)
        if st is not None and st["key"] != key:                # (a GAN / overlap switch never gets here: step() drops the graphs and warms up eagerly first)
            raise L.SmxError(f"TrainStep(use_graph): the captured step is for {st['key']}, got {key}; build another TrainStep for another shape")
        tf_new = transform if transform is not None else self._draw_transform(driving.shape[0], driving.device)
        if st is None:
            st = {"key": key, "src": source.float().contiguous().clone(), "drv": driving.float().contiguous().clone(), "tf": tf_new}
            g, g2 = torch.cuda.CUDAGraph(), (torch.cuda.CUDAGraph() if on_cut is not None else None)
            cap = {"ctx": torch.cuda.graph(g, capture_error_mode=CAPTURE_MODE)}

            def cut_capture():                                 # end graph 1 here, continue recording into graph 2 (same pool, replayed in order)
                cap["ctx"].__exit__(None, None, None)
                cap["ctx"] = torch.cuda.graph(g2, pool=g.pool(), capture_error_mode=CAPTURE_MODE)
                cap["ctx"].__enter__()
            cap["ctx"].__enter__()
            try:
                self.g.flat.zero_grad()
                self.flat_m.zero_grad()
                st["losses"], st["out"] = self.forward_backward(st["src"], st["drv"], w, st["tf"], gan, on_cut=cut_capture if g2 is not None else None)
            finally:
                cap["ctx"].__exit__(None, None, None)
            self._graph, self._graph2, self._static = g, g2, st      # (the capture executed nothing: the replay below is this step)
        else:
            st["src"].copy_(source)
            st["drv"].copy_(driving)
            if tf_new is not None:
                st["tf"].theta.copy_(tf_new.theta)
                if st["tf"].tps:
                    st["tf"].control_params.copy_(tf_new.control_params)
        self._graph.replay()
        if self._graph2 is not None:
            on_cut()
            self._graph2.replay()
        return st["losses"], st["out"]

    def step(self, source, driving, w=1.0, transform=None, ema=None, ema_decay=0.0, gan=False):
        """zero_grad -> forward/backward -> (all-reduce) -> Adam x2 -> (EMA).  With use_graph the returned loss / output tensors are the
        graph's static buffers: they are overwritten by the next step."""
        dist = self._dist()
        world = dist.get_world_size() if dist is not None else 1
        pending = []

        def start_g():                                         # net_g's gradients are final: their all-reduce runs under the estimator's backward
            pending.extend(self.g.flat.all_reduce_start(dist))
        on_cut = start_g if (dist is not None and self.overlap_allreduce) else None
        variant = (bool(gan), on_cut is not None)
        if variant != self._variant:
            # another step variant (the GAN branch switching on at net_d_start_iter, the overlap cut): its weight-gradient ReducePlan
            # has to RECORD first, and recording cannot happen inside a stream capture -- drop the captured graphs and run
            # GRAPH_WARMUP eager steps of the new variant before capturing again.  Plans of a variant that cannot come back
            # (gan=False once the branch is on) are released with their persistent workspaces.
            self._graph, self._graph2, self._static, self._eager_steps = None, None, None, 0
            if self._reduce_plans is not None and variant[0]:
                for k in [k for k in self._reduce_plans if isinstance(k, tuple) and not k[0]]:
                    del self._reduce_plans[k]
            self._variant = variant
        if self.use_graph and self._eager_steps >= self.GRAPH_WARMUP:
            losses, out = self._graph_step(source, driving, w, transform, gan, on_cut)
        else:
            self._eager_steps += 1
            self.g.flat.zero_grad()
            self.flat_m.zero_grad()
            losses, out = self.forward_backward(source, driving, w, transform, gan, on_cut)
        if dist is not None:
            if on_cut is None:
                start_g()
            pending.extend(self.flat_m.all_reduce_start(dist))
            FlatParams.all_reduce_wait(pending)
        self.g.flat.adam_step(self.g.lr, self.g.betas, self.g.eps, self.g.wd, gscale=1.0 / world)
        self.flat_m.adam_step(self.lr_m, self.betas_m, 1e-8, self.wd_m, gscale=1.0 / world)
        if ema is not None and ema_decay > 0:
            self.g.flat.ema_into(ema, ema_decay)
        if gan:                                                # optimize net_d (:408-430), after the generator / estimator update and the EMA
            self.flat_d.zero_grad()
            losses = dict(losses, **self.disc_backward(out["_out_nhwc"], out["_gt_nhwc"]))
            if world > 1:
                self.flat_d.all_reduce(dist)
            self.flat_d.adam_step(self.lr_d, self.betas_d, 1e-8, self.wd_d, gscale=1.0 / world)
        return losses, out
