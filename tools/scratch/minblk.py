import os, sys, time, torch, yaml
sys.path.insert(0, "/root/repo")
os.chdir("/root/repo")
from basicsr.archs import build_network
from synergize_motion_appearance_amd import driver, ops
from synergize_motion_appearance_amd.synth import synth_state_dict, synth_clip
cfg = yaml.safe_load(open("options/test.yml"))
net_g, me = build_network(cfg["network_g"]), build_network(cfg["network_motion_estimator"])
net_g.load_state_dict(synth_state_dict([(k, v.shape) for k, v in net_g.state_dict().items()]))
me.load_state_dict(synth_state_dict([(k, v.shape) for k, v in me.state_dict().items()]))
net_g, me = net_g.cuda().eval(), me.cuda().eval()
src, drv = synth_clip(300, seed=1)
drv = drv.cuda()
st = driver.encode_source_state(net_g, me, src.cuda()[None] if src.dim() == 3 else src.cuda(), drv[0:1], True)
for mb in (512, 256, 128, 64):
    ops.WINO_BF3_MIN_BLOCKS = mb
    row = []
    for b in (38, 75, 150):
        fr = [drv[(i * b) % (300 - b):(i * b) % (300 - b) + b] for i in range(10)]
        for f in fr[:2]: driver.render_frames(st, f, net_g, me, True, True, batch=b)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for f in fr[2:]: driver.render_frames(st, f, net_g, me, True, True, batch=b)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        row.append(f"B{b}: {8 * b / dt:7.1f} fps")
    print("min_blocks", mb, " | ".join(row), flush=True)
