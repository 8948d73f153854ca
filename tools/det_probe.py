import sys, os, torch, yaml
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from basicsr.archs import build_network
from synergize_motion_appearance_amd import driver, ops
from synergize_motion_appearance_amd.synth import synth_state_dict, synth_clip
cfg = yaml.safe_load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "options/test.yml")))
net_g, me = build_network(cfg["network_g"]), build_network(cfg["network_motion_estimator"])
net_g.load_state_dict(synth_state_dict([(k, v.shape) for k, v in net_g.state_dict().items()]), strict=True)
me.load_state_dict(synth_state_dict([(k, v.shape) for k, v in me.state_dict().items()]), strict=True)
net_g, me = net_g.cuda().eval(), me.cuda().eval()
src, drv = synth_clip(11, seed=9)
u8 = ops.to_uint8(drv.cuda().permute(0, 2, 3, 1).contiguous(), -1.0, 1.0).cpu()
x = ops.frames_u8_to_nchw(u8.cuda())
st = driver.encode_source_state(net_g, me, src.cuda(), x[0:1], True)
a = driver.render_frames(st, x, net_g, me, True, True, batch=4)
one = torch.cat([driver.render_frames(st, x[i:i + 1], net_g, me, True, True, batch=1) for i in range(11)])
print("render(b4) vs per-frame:", [(a[i].int() - one[i].int()).abs().max().item() for i in range(11)])
pipe = driver.FramePipeline(net_g, me, batch=4)
for trial in range(3):
    c = pipe.run(st, u8)
    print("pipe vs render:", [(c[i].int() - a[i].cpu().int()).abs().max().item() for i in range(11)])
# same with a device sync after every H2D (is it the copy / event ordering?)

torch.cuda.synchronize()

