"""SURVEY row N2 slice 2 / BASELINE configs[4]: the generator-side TRAINING STEP on the HIP path against the reference's own
`l_g_total.backward()` (fixture `tests/golden/train_step_netg.npz`, produced by `make_golden_r3.py` from the imported reference:
AppMotionCompFormer.train(), B=2 (source, driving) pairs, L1 pixel + motion / appearance codebook + motion reconstruction +
low-resolution pixel losses, then one torch.optim.Adam step with options/train.yml's settings).

Bars: losses 1e-4 relative; the gradient NORM of each of the 472 parameters within 1e-3 relative (plus an absolute floor of 1e-6 of
the largest norm for gradients that are analytically zero: a bias in front of a one-channel-per-group GroupNorm, the key bias of a
softmax); stored gradient tensors within 1e-3 of their own max; gradients w.r.t. the dense-motion inputs within 2e-3; Adam's parameter update within 2 % of the step size where |g| is not at the rounding floor."""
import os

import numpy as np
import pytest
import torch
import yaml

from tests.util import golden, weights, HERE

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(HERE)


@pytest.fixture(scope="module")
def setup():
    assert torch.cuda.is_available(), "needs an MI355X"
    from basicsr.archs import build_network
    from synergize_motion_appearance_amd.synth import synth_clip
    from synergize_motion_appearance_amd.trainer import NetGTrainStep
    cfg = yaml.safe_load(open(os.path.join(REPO, "options/train.yml")))
    net_g = build_network(cfg["network_g"])
    net_g.load_state_dict(weights("network_g"), strict=True)
    net_g = net_g.cuda()
    g = golden("train_step_netg.npz")
    _, clip = synth_clip(8, seed=int(g["clip_seed"]))
    src, drv = clip[g["src_frames"].tolist()].contiguous().cuda(), clip[g["drv_frames"].tolist()].contiguous().cuda()
    dm = {k: torch.from_numpy(g["in_" + k]).cuda() for k in ("deformation", "occlusion_map", "driving_kp_heatmap")}
    train_opt = {k: v for k, v in cfg["train"].items() if k not in ("perceptual_opt", "gan_opt", "equivariance_opt", "kp_distance_opt")}
    step = NetGTrainStep(net_g, train_opt)
    return g, step, src, drv, dm


def _sub(t, s0, s1):
    return t[::s0] if t.dim() == 1 else t.reshape(t.shape[0], -1)[::s0, ::s1]


def test_generator_step_losses_and_gradients_vs_reference(setup):
    from tests.golden.make_golden_r3 import SAMPLES
    g, step, src, drv, dm = setup
    step.flat.zero_grad()
    losses, out, gin = step.forward_backward(src, drv, dm)
    torch.cuda.synchronize()
    assert float((out["out"].cpu()[:, :, ::4, ::4] - torch.from_numpy(g["out"])).abs().max()) < 1e-3
    assert float((out["out_lr"][0].cpu()[:, :, ::4, ::4] - torch.from_numpy(g["out_lr"])).abs().max()) < 1e-3
    for k in ("l_g_pix", "l_g_motion_codebook_code", "l_g_motion_codebook_recon", "l_g_pix_lr_0", "l_g_app_codebook_code"):
        ref = float(g["loss_" + k])
        assert abs(float(losses[k]) - ref) < 2e-4 * abs(ref), (k, float(losses[k]), ref)
    assert abs(float(losses["l_g_total"]) - float(g["l_g_total"])) < 2e-4 * float(g["l_g_total"])
    # every parameter's gradient norm
    names = [str(n) for n in g["param_names"]]
    ref_norm = g["grad_norms"]
    assert names == list(step.flat.G)
    mine = np.array([float(step.flat.G[n].double().norm()) for n in names])
    floor = 1e-6 * ref_norm.max()
    bad = [(n, a, b) for n, a, b in zip(names, mine, ref_norm) if abs(a - b) > 1e-3 * b + floor]
    assert not bad, bad[:10]
    # stored gradient tensors
    for n, s0, s1 in SAMPLES:
        ref = torch.from_numpy(g["grad:" + n])
        got = _sub(step.flat.G[n], s0, s1).cpu()
        assert got.shape == ref.shape, n
        # 1e-3 of the tensor's largest entry; gradients that are themselves at the 1e-6 level (position_emb_motion: a sum of
        # cancelling contributions of order 1e-3) sit on fp32 rounding of their summands: 5e-3 there
        mx = float(ref.abs().max())
        assert float((got - ref).abs().max()) < (1e-3 if mx >= 1e-5 else 5e-3) * mx, (n, float((got - ref).abs().max()), mx)
    # gradients w.r.t. the dense-motion inputs (what the motion estimator's backward receives)
    for k in ("deformation", "occlusion_map", "driving_kp_heatmap"):
        ref = torch.from_numpy(g["grad_in_" + k])
        assert float((gin[k].cpu() - ref).abs().max()) < 2e-3 * float(ref.abs().max()), k      # measured 4e-4 / 5e-4 / 1.1e-3


def test_generator_step_adam_update_vs_reference(setup):
    from tests.golden.make_golden_r3 import ADAM_SAMPLES
    g, step, src, drv, dm = setup
    before = {n: step.flat.P[n].detach().clone() for n in ADAM_SAMPLES}
    losses, _, _ = step.step(src, drv, dm)
    torch.cuda.synchronize()
    for n in ADAM_SAMPLES:
        ref = torch.from_numpy(g["adam_delta:" + n])
        got = (step.flat.P[n] - before[n]).reshape(-1)[:4096].cpu()
        # first Adam step: |delta| = lr * |g| / (|g| + eps) ~= lr = 8e-5 with the sign of -g
        big = ref.abs() > 4e-5
        assert float((got[big] - ref[big]).abs().max()) < 2e-6, n
    # the module's parameters ARE the flat slots: state_dict reflects the update
    sd = step.net_g.state_dict()
    assert torch.equal(sd["generator.blocks.18.weight"], step.flat.P["generator.blocks.18.weight"])
    # a second step runs and lowers nothing catastrophically (finite losses)
    l2, _, _ = step.step(src, drv, dm)
    assert all(torch.isfinite(v).all() for v in l2.values())
