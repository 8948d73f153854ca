import os, sys, numpy as np, torch, yaml, time
sys.path.insert(0, "/root/repo")
from tests.util import golden, weights
from tests.golden.make_golden_r3 import SAMPLES, ADAM_SAMPLES
from basicsr.archs import build_network
from synergize_motion_appearance_amd.synth import synth_clip
from synergize_motion_appearance_amd.trainer import NetGTrainStep
cfg = yaml.safe_load(open("/root/repo/options/train.yml"))
net_g = build_network(cfg["network_g"]); net_g.load_state_dict(weights("network_g"), strict=True); net_g = net_g.cuda()
g = golden("train_step_netg.npz")
_, clip = synth_clip(8, seed=int(g["clip_seed"]))
src, drv = clip[g["src_frames"].tolist()].contiguous().cuda(), clip[g["drv_frames"].tolist()].contiguous().cuda()
dm = {k: torch.from_numpy(g["in_" + k]).cuda() for k in ("deformation", "occlusion_map", "driving_kp_heatmap")}
train_opt = {k: v for k, v in cfg["train"].items() if k not in ("perceptual_opt", "gan_opt", "equivariance_opt", "kp_distance_opt")}
step = NetGTrainStep(net_g, train_opt)
step.flat.zero_grad()
losses, out, gin = step.forward_backward(src, drv, dm)
torch.cuda.synchronize()
sub = lambda t, s0, s1: t[::s0] if t.dim() == 1 else t.reshape(t.shape[0], -1)[::s0, ::s1]
rows = []
for n, s0, s1 in SAMPLES:
    ref = torch.from_numpy(g["grad:" + n]); got = sub(step.flat.G[n], s0, s1).cpu()
    rows.append((float((got - ref).abs().max() / ref.abs().max()), float(ref.abs().max()), n))
for r in sorted(rows, reverse=True)[:12]: print("%.2e  max|g| %.2e  %s" % r)
names = [str(n) for n in g["param_names"]]; rn = g["grad_norms"]
mine = np.array([float(step.flat.G[n].double().norm()) for n in names])
e = np.abs(mine - rn) / rn
o = np.argsort(-e)[:8]
print("worst norm rel errs:", [(names[i], float(e[i]), float(rn[i])) for i in o])
for k in gin: 
    ref = torch.from_numpy(g["grad_in_" + k]); print(k, float((gin[k].cpu() - ref).abs().max() / ref.abs().max()))
# vq index agreement is implicit in codebook grads; timing of a step
for _ in range(2): step.step(src, drv, dm)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(3): step.step(src, drv, dm)
torch.cuda.synchronize(); print("ms/step B=2:", (time.perf_counter() - t0) / 3 * 1e3)
