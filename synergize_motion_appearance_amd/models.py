"""MODEL-level entry of the animation path: `build_model(opt)` and
the inference surface of `AppMotionCompModel` -- `feed_data`, `test`, `make_animation`,
`generate_video_image`, `load_network` -- as `basicsr/animate.py:73-78` and `basicsr/test.py:75-80`
drive them (reference `basicsr/models/__init__.py:19-30`, `sr_model.py:17-31`,
`base_model.py:17-22, 236-262`, `appmotioncomp_model.py:110-114, 437-456, 607-756`).

Same names, argument meaning and return types (BGR uint8 HWC arrays, the same files at the same
relative paths); the per-frame loop inside is the batched HIP path of `driver.animate_batched`:
one keypoint / dense-motion / generator launch sequence per `val.batch` frames, the source
encoded once per video.

Training members (row N2, BASELINE configs[4]; `appmotioncomp_model.py:116-434, 598-605`): `init_training_settings`,
`optimize_parameters`, `model_ema`, `update_learning_rate`, `save` drive `trainer.TrainStep` -- the reference step (generator + motion
estimator, and past `net_d_start_iter` the hinge-GAN branch with its adaptive weight and the discriminator's own update) as HIP kernels
with a gradient tape, Adam per network on flat parameter buffers, gradients summed over `torch.distributed` (RCCL) when it is
initialised.  The term that needs weights the reference downloads is NOT silently dropped: `perceptual_opt` (VGG19) raises unless
`perceptual_opt.vgg19_path` is given or `train.allow_missing_losses` declares the skip -- and the GAN branch, whose adaptive weight is
measured against the perceptual loss, raises when reached without it."""
from collections import OrderedDict
from copy import deepcopy
from os import path as osp

import numpy as np
import torch

from . import driver, ops
from .archs import build_network
from .img_util import imwrite, mimsave, tensor2img
from .registry import MODEL_REGISTRY

__all__ = ["build_model", "AppMotionCompModel", "calculate_psnr", "calculate_l1", "calculate_ssim"]


def build_model(opt):
    """reference `basicsr/models/__init__.py:19-30`: deep copy, `opt['model_type']` -> class(opt)."""
    opt = deepcopy(opt)
    return MODEL_REGISTRY.get(opt["model_type"])(opt)


# ---- the three array metrics of `val.metrics` that need no external network ----------------------
def _crop(img, crop_border):
    return img if crop_border == 0 else img[crop_border:-crop_border, crop_border:-crop_border, ...]


def calculate_psnr(img1, img2, crop_border=0, test_y_channel=False, **_):
    """`basicsr/metrics/psnr_ssim.py` calculate_psnr on uint8/[0,255] HWC arrays."""
    if test_y_channel:
        raise NotImplementedError("test_y_channel needs the reference's bgr2ycbcr tables; the shipped yml uses false")
    a, b = _crop(np.asarray(img1, np.float64), crop_border), _crop(np.asarray(img2, np.float64), crop_border)
    mse = np.mean((a - b) ** 2)
    return float("inf") if mse == 0 else float(20.0 * np.log10(255.0 / np.sqrt(mse)))


def calculate_l1(img1, img2, crop_border=0, **_):
    a, b = _crop(np.asarray(img1, np.float64), crop_border), _crop(np.asarray(img2, np.float64), crop_border)
    return float(np.mean(np.abs(a - b)))


def calculate_ssim(img1, img2, crop_border=0, test_y_channel=False, **_):
    """SSIM with the 11x11 sigma-1.5 Gaussian window, 'valid' region, averaged over channels."""
    if test_y_channel:
        raise NotImplementedError("test_y_channel needs the reference's bgr2ycbcr tables")
    from scipy.signal import convolve2d
    a, b = _crop(np.asarray(img1, np.float64), crop_border), _crop(np.asarray(img2, np.float64), crop_border)
    g = np.exp(-((np.arange(11) - 5.0) ** 2) / (2 * 1.5 ** 2))
    g /= g.sum()
    win = np.outer(g, g)
    c1, c2 = (0.01 * 255) ** 2, (0.03 * 255) ** 2

    def one(x, y):
        f = lambda z: convolve2d(z, win, mode="valid")   # noqa: E731
        mx, my = f(x), f(y)
        sxx, syy, sxy = f(x * x) - mx * mx, f(y * y) - my * my, f(x * y) - mx * my
        return float((((2 * mx * my + c1) * (2 * sxy + c2)) / ((mx * mx + my * my + c1) * (sxx + syy + c2))).mean())
    if a.ndim == 2:
        return one(a, b)
    return float(np.mean([one(a[..., i], b[..., i]) for i in range(a.shape[2])]))


_ARRAY_METRICS = {"calculate_psnr": calculate_psnr, "calculate_l1": calculate_l1, "calculate_ssim": calculate_ssim}


@MODEL_REGISTRY.register()
class AppMotionCompModel:
    """The reference model class (`appmotioncomp_model.py:108`): inference surface, and -- with opt['is_train'] -- the generator /
    motion-estimator training step (see the module docstring for what is deliberately absent)."""

    def __init__(self, opt):
        self.opt = opt
        if opt.get("num_gpu", 1) == 0:
            raise RuntimeError("the MI355X-native path has no CPU mode (num_gpu: 0)")
        self.device = torch.device("cuda")
        self.is_train = opt.get("is_train", False)
        self.net_g = build_network(opt["network_g"]).to(self.device).eval()
        path = opt.get("path", {})
        if path.get("pretrain_network_g") is not None:
            self.load_network(self.net_g, path["pretrain_network_g"], path.get("strict_load_g", True),
                              path.get("param_key_g", "params"))
        self.motion_estimator = None
        self.metric_results = {}
        if self.is_train:
            self.init_training_settings()

    # -- appmotioncomp_model.py:116-221 (what of it the HIP step consumes) ------------------------------------------------------
    def init_training_settings(self):
        from .trainer import TrainStep, FlatParams
        train_opt = self.opt["train"]
        # perceptual_opt (MultiScalePyramidPerceptualLoss) is built when it can get its VGG19 weights (perceptual_opt.vgg19_path, or
        # synthetic_vgg19 for benchmark runs); without them: an error, or -- train.allow_missing_losses: true -- a declared skip
        po = train_opt.get("perceptual_opt")
        has_vgg = bool(po) and bool(dict(po).get("vgg19_path") or dict(po).get("synthetic_vgg19"))
        missing = ["perceptual_opt"] if (po and not has_vgg) else []
        if missing and not train_opt.get("allow_missing_losses", False):
            raise NotImplementedError("train.perceptual_opt (MultiScalePyramidPerceptualLoss) needs the torchvision VGG19 weights the reference "
                                      "downloads (archs/vgg_arch.py:173): set perceptual_opt.vgg19_path, or train.allow_missing_losses: true to train "
                                      "without that term")
        self.skipped_losses = missing
        self.ema_decay = float(train_opt.get("ema_decay", 0) or 0)
        me = self._ensure_motion_estimator()
        self.net_g.train()
        me.train()
        # the discriminator (appmotioncomp_model.py:137-160): built when the yml has network_d and train.gan_opt; its branch switches on past
        # net_d_start_iter (adaptive weight = perceptual + pixel vs GAN gradient: it needs the perceptual term like the reference does)
        self.net_d = None
        if self.opt.get("network_d") and train_opt.get("gan_opt"):
            self.net_d = build_network(self.opt["network_d"]).to(self.device)
            path = self.opt.get("path", {})
            if path.get("pretrain_network_d") is not None:
                self.load_network(self.net_d, path["pretrain_network_d"], path.get("strict_load_d", True), "params_d")
            self.net_d.train()
        self.train_step = TrainStep(self.net_g, me, {k: v for k, v in train_opt.items() if (k != "perceptual_opt" or has_vgg)}, net_d=self.net_d)
        self.net_g_ema, self._ema_flat = None, None
        if self.ema_decay > 0:
            self.net_g_ema = build_network(self.opt["network_g"]).to(self.device).eval()
            self._ema_flat = FlatParams(self.net_g_ema)
            path = self.opt.get("path", {})
            if path.get("pretrain_network_g") is not None:
                self.load_network(self.net_g_ema, path["pretrain_network_g"], path.get("strict_load_g", True), "params_ema")
            else:
                self.model_ema(0)
        self.net_g_start_iter = train_opt.get("net_g_start_iter", 0)
        self.net_d_iters = train_opt.get("net_d_iters", 1)
        self.net_d_start_iter = train_opt.get("net_d_start_iter", 0)
        sch = dict(train_opt.get("scheduler") or {})
        self._milestones, self._gamma = list(sch.get("milestones", [])), float(sch.get("gamma", 1.0))
        # one base rate per optimizer, the reference's order [g, m (, d)] (appmotioncomp_model.py:248, 265, 270): MultiStepLR decays all three
        self._base_lr = [self.train_step.g.lr, self.train_step.lr_m] + ([self.train_step.lr_d] if self.net_d is not None else [])
        self.log_dict = OrderedDict()

    def model_ema(self, decay=0.999):
        """models/sr_model.py model_ema: net_g_ema = decay * net_g_ema + (1 - decay) * net_g (one kernel over the flat buffers)."""
        self.train_step.g.flat.ema_into(self._ema_flat, decay)
        self.net_g_ema.refresh()

    def update_learning_rate(self, current_iter, warmup_iter=-1):
        """base_model.py:144-165 for the shipped MultiStepLR: lr = base * gamma^(milestones passed), linear warm-up when configured."""
        k = sum(1 for m in self._milestones if current_iter > m)
        f = self._gamma ** k
        if 0 < current_iter < warmup_iter:
            f *= current_iter / warmup_iter
        self.train_step.g.lr, self.train_step.lr_m = self._base_lr[0] * f, self._base_lr[1] * f
        if self.net_d is not None:
            self.train_step.lr_d = self._base_lr[2] * f

    def get_current_learning_rate(self):
        """base_model.py:167-168: the FIRST optimizer's param-group rates (what train.py logs)."""
        return [self.train_step.g.lr]

    def get_current_learning_rates(self):
        """one rate per optimizer, the reference's order [g, m (, d)]."""
        return [self.train_step.g.lr, self.train_step.lr_m] + ([self.train_step.lr_d] if self.net_d is not None else [])

    def get_current_log(self):
        return self.log_dict

    def optimize_parameters(self, current_iter):
        """appmotioncomp_model.py:294-434: motion_estimator(gt, source) -> net_g(source, dense_motion, w=1, gt=gt) -> l_g_total.backward()
        -> optimizer_g.step(), optimizer_m.step() -> EMA; then `reduce_loss_dict`."""
        if not self.is_train:
            raise RuntimeError("optimize_parameters needs opt['is_train'] = True")
        gan = current_iter > self.net_d_start_iter
        if gan and (self.net_d is None or self.train_step.percep is None):
            raise RuntimeError(f"iteration {current_iter} is past net_d_start_iter = {self.net_d_start_iter}: the GAN branch (appmotioncomp_model.py:324-345, "
                               "408-432) needs network_d + train.gan_opt and the perceptual loss its adaptive weight is measured against "
                               "(perceptual_opt.vgg19_path)")
        loss_dict = OrderedDict()
        if current_iter % self.net_d_iters == 0 and current_iter > self.net_g_start_iter:
            losses, self.out_dict = self.train_step.step(self.source, self.gt, w=1.0, gan=gan)
            self.dense_motion = {k: self.out_dict[k] for k in ("deformation", "occlusion_map", "kp_driving", "kp_source")}
            for k, v in losses.items():
                if k != "l_g_total" and not k.startswith("_"):
                    loss_dict[k] = v.detach().reshape(())
        if self.ema_decay > 0:
            self.model_ema(decay=self.ema_decay)
        self.log_dict = self.reduce_loss_dict(loss_dict)

    def reduce_loss_dict(self, loss_dict):
        """base_model.py:298-323: average over ranks (to rank 0), then python floats."""
        import torch.distributed as dist
        with torch.no_grad():
            if loss_dict and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
                keys = list(loss_dict)
                t = torch.stack([loss_dict[k] for k in keys], 0)
                if dist.get_backend() == "gloo":
                    h = t.cpu()
                    dist.reduce(h, dst=0)
                    t = h
                else:
                    dist.reduce(t, dst=0)
                if dist.get_rank() == 0:
                    t = t / dist.get_world_size()
                loss_dict = OrderedDict(zip(keys, t))
            return OrderedDict((k, float(v)) for k, v in loss_dict.items())

    def save_network(self, net, net_label, current_iter, param_key="params"):
        """base_model.py:171-200: `{net_label}_{iter}.pth` holding {'params': state_dict} (and 'params_ema')."""
        import os
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_rank() != 0:
            return
        nets = net if isinstance(net, list) else [net]
        keys = param_key if isinstance(param_key, list) else [param_key]
        name = f"{net_label}_{'latest' if current_iter == -1 else current_iter}.pth"
        os.makedirs(self.opt["path"]["models"], exist_ok=True)
        torch.save({k: OrderedDict((n, v.detach().cpu().clone()) for n, v in n_.state_dict().items()) for n_, k in zip(nets, keys)},
                   os.path.join(self.opt["path"]["models"], name))

    # -- base_model.py:236-262 ------------------------------------------------------------------------
    def load_network(self, net, load_path, strict=True, param_key="params"):
        load_net = torch.load(load_path, map_location="cpu")
        if param_key is not None:
            if param_key not in load_net and "params" in load_net:
                param_key = "params"
            load_net = load_net[param_key]
        load_net = OrderedDict((k[7:] if k.startswith("module.") else k, v) for k, v in load_net.items())
        net.load_state_dict(load_net, strict=strict)

    def _ensure_motion_estimator(self):
        """appmotioncomp_model.py:658-668 (built, and its checkpoint loaded, at validation time)."""
        if self.motion_estimator is None:
            if self.opt.get("network_motion_estimator") is None:
                raise KeyError("opt['network_motion_estimator'] is required by the animation path")
            me = build_network(self.opt["network_motion_estimator"]).to(self.device).eval()
            path = self.opt.get("path", {})
            if path.get("pretrain_network_motion_estimator") is not None:
                self.load_network(me, path["pretrain_network_motion_estimator"],
                                  path.get("strict_load_motion_estimator", True), path.get("param_key_m", "params"))
            self.motion_estimator = me
        return self.motion_estimator

    # -- appmotioncomp_model.py:110-114, 437-456 --------------------------------------------------------
    def feed_data(self, data):
        self.gt = data["driving"].to(self.device)
        self.source = data["source"].to(self.device)
        self.b = self.gt.shape[0]

    @torch.no_grad()
    def test(self):
        me = self._ensure_motion_estimator()
        if self.is_train:                     # parameters moved since the inference engines packed them
            self.net_g.refresh()
            me.refresh()
        self.dense_motion = me(self.gt, self.source)
        self.out_dict = self.net_g(self.source, self.dense_motion, w=self.opt.get("val", {}).get("w", 1), inference=True)
        self.driving_feat = self.net_g.encode_driving(self.gt)
        self.source_feat = self.net_g.encode_driving(self.source)
        if "lq_feat" in self.out_dict:
            self.lq_recon = self.net_g.generator(self.out_dict["lq_feat"])

    def get_current_visuals(self):
        out = OrderedDict()
        out["gt"], out["source"] = self.gt.detach().cpu(), self.source.detach().cpu()
        out["result"] = self.out_dict["out"].detach().cpu()
        if hasattr(self, "lq_recon"):                        # generator(lq_feat): the un-fused decode (appmotioncomp_model.py:590-592)
            out["recon"] = self.lq_recon.detach().cpu()
        return out

    # -- appmotioncomp_model.py:607-639 -----------------------------------------------------------------
    @torch.no_grad()
    def make_animation(self, source_img, driving, cpu=False, anchor_idx=0):
        """source [1,3,H,W]; driving: list of [1,3,H,W] (or [N,3,H,W]) in [-1,1] ->
        (predictions, driving_imgs): lists of BGR uint8 HWC arrays, one per driving frame.
        `driving[anchor_idx]` supplies kp_driving_initial (the reference always uses index 0 and gets
        the effect of an anchor by reversing lists; see `generate_video_image`)."""
        if cpu:
            raise RuntimeError("the MI355X-native path has no CPU mode")
        val = self.opt.get("val", {})
        relative, adapt = val.get("relative", False), val.get("adapt_scale", False)
        me = self._ensure_motion_estimator()
        drv = driving if torch.is_tensor(driving) else torch.cat([d if d.dim() == 4 else d[None] for d in driving])
        self.gt, self.source = driving, source_img.to(self.device)
        self.b = 1
        drv = drv.to(self.device).float()
        out = driver.animate_batched(self.source.float(), drv, self.net_g, me, relative=relative,
                                     adapt_movement_scale=adapt, batch=int(val.get("batch", 30)), anchor_idx=anchor_idx,
                                     w=float(val.get("w", 1)))
        preds = [np.ascontiguousarray(p[:, :, ::-1]) for p in out.cpu().numpy()]
        d8 = ops.to_uint8(ops.nchw_to_nhwc(drv), -1.0, 1.0).cpu().numpy()
        return preds, [np.ascontiguousarray(d[:, :, ::-1]) for d in d8]

    # -- appmotioncomp_model.py:642-756 -----------------------------------------------------------------
    @torch.no_grad()
    def generate_video_image(self, dataloader, current_iter, tb_logger):
        dataset_name = dataloader.dataset.opt["name"]
        val = self.opt.get("val", {})
        metrics = val.get("metrics")
        with_metrics = metrics is not None
        if with_metrics:
            self.metric_results = {m: 0 for m in metrics.keys()}
        vis_root = osp.join(self.opt["path"]["visualization"], dataset_name)
        self.net_g.eval()
        self._ensure_motion_estimator()
        count_num = 0
        for val_data in dataloader:
            video_name, anchor_idx = val_data["video_name"], val_data["anchor_idx"]
            driving_video = val_data["driving_video"]
            if torch.is_tensor(anchor_idx):
                anchor_idx = int(anchor_idx)
            if anchor_idx is None:
                raise TypeError("val_data['anchor_idx'] is None (the reference fails here too: "
                                "driving_forward is unbound, appmotioncomp_model.py:679-683)")
            # reference: make_animation(driving[a:]) ++ make_animation(driving[:a+1][::-1]) spliced as
            # backward[::-1] + forward[1:]; both halves start at driving[a], so kp_driving_initial is
            # kp(driving[a]) for every frame -> one pass over the frames in order with that anchor.
            predictions, drivings = self.make_animation(val_data["source"], driving_video, anchor_idx=anchor_idx)
            source = tensor2img([self.source.detach().cpu()], rgb2bgr=True, min_max=(-1, 1))
            visual = []
            for i in range(len(predictions)):
                vis = np.concatenate((source, drivings[i], predictions[i]), axis=1)
                visual.append(vis)
                img_name = video_name[0] + "_" + val_data["driving_name_list"][i][0]
                imwrite(vis, osp.join(vis_root, "visual", f"{img_name}_v.png"))
                imwrite(predictions[i], osp.join(vis_root, "result", f"{img_name}_r.png"))
                imwrite(source, osp.join(vis_root, "source", f"{img_name}_s.png"))
                imwrite(drivings[i], osp.join(vis_root, "driving", f"{img_name}_d.png"))
                if with_metrics:
                    for name, opt_ in metrics.items():
                        if name in ("psnr", "ssim", "l1"):
                            o = dict(opt_)
                            fn = _ARRAY_METRICS[o.pop("type")]
                            self.metric_results[name] += fn(predictions[i], drivings[i], **o)
                            count_num += 1
            mimsave([np.ascontiguousarray(p[:, :, ::-1]) for p in predictions],
                    osp.join(vis_root, "result_videos", f"{video_name[0]}_r.mp4"))
            mimsave([np.ascontiguousarray(v[:, :, ::-1]) for v in visual],
                    osp.join(vis_root, "visual_videos", f"{video_name[0]}_v.mp4"))
        if with_metrics:
            for metric in list(metrics.keys()):
                if metric in ("psnr", "ssim", "l1"):
                    # the reference divides by the count accumulated over ALL array metrics
                    # (appmotioncomp_model.py:706-711, 722-724): kept, so the logged numbers agree
                    self.metric_results[metric] /= max(count_num, 1)
                    if metric == "l1":
                        self.metric_results["l1_255"] = self.metric_results["l1"] / 255.0
                else:
                    # fid / lpips / face_akd / face_aed / id_similarity / pose_accuracy need networks the
                    # reference downloads (InceptionV3, VGG, face_alignment, ArcFace, hopenet): not built
                    self.metric_results[metric] = float("nan")
        return self.metric_results

    def validation(self, dataloader, current_iter, tb_logger, save_img=False, **kwargs):
        """base_model.py:39-52: distributed validation runs on rank 0 only (`dist_validation`, appmotioncomp_model.py:458-460)."""
        if self.opt.get("dist") and self.opt.get("rank", 0) != 0:
            return None
        return self.nondist_validation(dataloader, current_iter, tb_logger, save_img)

    @torch.no_grad()
    def nondist_validation(self, dataloader, current_iter, tb_logger, save_img):
        """appmotioncomp_model.py:463-566 -- the frame-pair evaluation `test.py` drives: per item feed_data -> test() -> the four PNGs
        (`visual/_v`, `result/_r`, `source/_s`, `driving/_d` under path.visualization/<dataset>) and the array metrics psnr / ssim / l1
        (averaged over the items; metrics that need downloaded networks are NaN, as in generate_video_image)."""
        dataset_name = dataloader.dataset.opt["name"]
        metrics = self.opt.get("val", {}).get("metrics")
        self.metric_results = {m: 0 for m in metrics.keys()} if metrics is not None else {}
        self._ensure_motion_estimator()
        vis_root = osp.join(self.opt["path"]["visualization"], dataset_name)
        n = 0
        for val_data in dataloader:
            img_name = val_data["frame_name"][0]
            self.feed_data(val_data)
            self.test()
            visuals = self.get_current_visuals()
            result_img = tensor2img([visuals["result"]], rgb2bgr=True, min_max=(-1, 1))
            gt_img = tensor2img([visuals["gt"]], rgb2bgr=True, min_max=(-1, 1))
            source = tensor2img([visuals["source"]], rgb2bgr=True, min_max=(-1, 1))
            visual = tensor2img([torch.cat((visuals["source"], visuals["gt"], visuals["result"]), 3)], rgb2bgr=True, min_max=(-1, 1))
            if "recon" in visuals:
                visual = np.concatenate((visual, tensor2img(visuals["recon"], rgb2bgr=True, min_max=(-1, 1))), axis=1)
            if save_img:
                if self.opt.get("is_train"):
                    imwrite(visual, osp.join(self.opt["path"]["visualization"], img_name, f"{img_name}_{current_iter}.png"))
                else:
                    imwrite(visual, osp.join(vis_root, "visual", f"{img_name}_v.png"))
                    imwrite(result_img, osp.join(vis_root, "result", f"{img_name}_r.png"))
                    imwrite(source, osp.join(vis_root, "source", f"{img_name}_s.png"))
                    imwrite(gt_img, osp.join(vis_root, "driving", f"{img_name}_d.png"))
            if metrics is not None:
                for name, opt_ in metrics.items():
                    if name in ("psnr", "ssim", "l1"):
                        o = dict(opt_)
                        self.metric_results[name] += _ARRAY_METRICS[o.pop("type")](result_img, gt_img, **o)
            n += 1
        if metrics is not None:
            for metric in list(metrics.keys()):
                if metric in ("psnr", "ssim", "l1"):
                    self.metric_results[metric] /= max(n, 1)
                    if metric == "l1":
                        self.metric_results["l1_255"] = self.metric_results["l1"] / 255.0
                else:
                    self.metric_results[metric] = float("nan")
        return self.metric_results

    def save(self, epoch, current_iter):
        """appmotioncomp_model.py:598-605 (the optimizer / scheduler state file is the harness side: not written)."""
        if self.ema_decay > 0:
            self.save_network([self.net_g, self.net_g_ema], "net_g", current_iter, param_key=["params", "params_ema"])
        else:
            self.save_network(self.net_g, "net_g", current_iter)
        self.save_network(self.motion_estimator, "net_motion_estimator", current_iter)
        if getattr(self, "net_d", None) is not None:
            self.save_network(self.net_d, "net_d", current_iter)
        self.save_training_state(epoch, current_iter)

    def save_training_state(self, epoch, current_iter):
        """base_model.py:265-281: `{iter}.state` = {'epoch', 'iter', 'optimizers': [g, m, d], 'schedulers': [...]}, the optimizer entries in
        torch.optim.Adam's own state_dict layout (the flat Adam buffers sliced back per parameter), so the file is interchangeable with the
        reference's."""
        if current_iter == -1:
            return
        import os
        sched = {"milestones": list(self._milestones), "gamma": self._gamma, "last_epoch": int(current_iter)}
        state = {"epoch": epoch, "iter": current_iter, "optimizers": self.train_step.optimizer_state_dicts(),
                 "schedulers": [dict(sched, base_lrs=[lr]) for lr in self._base_lr]}
        d = self.opt["path"].get("training_states") or os.path.join(self.opt["path"].get("models", "."), "..", "training_states")
        os.makedirs(d, exist_ok=True)
        torch.save(state, os.path.join(d, f"{current_iter}.state"))

    def resume_training(self, resume_state):
        """base_model.py:283-296: reload the Adam moments / step counts; the learning rate follows `update_learning_rate(current_iter)`."""
        self.train_step.load_optimizer_state_dicts(resume_state["optimizers"])
        self.train_step.sync_replicas()                      # multi-rank: every replica continues from rank 0's parameters / moments / buffers
