"""Stage-level and end-to-end parity of the HIP path on a real MI355X, through the drop-in
plugin surface (build_network / ARCH_REGISTRY names / strict checkpoint load), against the
fixtures generated from the imported reference (tests/golden) and against the CPU oracle.
Bar (BASELINE.json north_star): <= 1e-3 max-abs on fp32 pixels."""
import os

import numpy as np
import pytest
import torch
import yaml

from oracle import reenact_oracle as O
from tests.util import golden, weights, clip, maxabs, HERE
from synergize_motion_appearance_amd.synth import synth_keypoints

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(HERE)


@pytest.fixture(scope="module")
def nets():
    assert torch.cuda.is_available(), "needs an MI355X"
    from basicsr.archs import build_network
    from basicsr.utils.options import ordered_yaml
    cfg = yaml.load(open(os.path.join(REPO, "options/test.yml")), Loader=ordered_yaml()[0])
    net_g = build_network(cfg["network_g"])
    me = build_network(cfg["network_motion_estimator"])
    net_g.load_state_dict(weights("network_g"), strict=True)
    me.load_state_dict(weights("network_motion_estimator"), strict=True)
    return net_g.eval().cuda(), me.eval().cuda()


def _cuda(d):
    return {k: v.cuda() for k, v in d.items()}


def _kp(g, which, idx=None):
    v, j = g[f"{which}_value"], g[f"{which}_jacobian"]
    if idx is not None:
        v, j = v[idx], j[idx]
    return {"value": torch.from_numpy(v).cuda(), "jacobian": torch.from_numpy(j).cuda()}


def test_estimate_kp_vs_reference(nets):
    _, me = nets
    src, drv = clip()
    g = golden("kp.npz")
    s = me.estimate_kp(src[None].cuda())
    d = me.estimate_kp(drv.cuda())                      # B = 8 frames in one launch
    assert maxabs(s["value"].cpu(), g["src_value"]) < 1e-5 and maxabs(s["jacobian"].cpu(), g["src_jacobian"]) < 2e-5
    assert maxabs(d["value"].cpu(), g["drv_value"]) < 1e-5 and maxabs(d["jacobian"].cpu(), g["drv_jacobian"]) < 2e-5


def test_dense_motion_vs_reference(nets):
    _, me = nets
    src, _ = clip()
    g, gk = golden("dense_motion.npz"), golden("kp.npz")
    dm = me.estimate_motion_w_kp(kp_source=_kp(gk, "src"), kp_driving=_kp(gk, "drv", [2, 5]), source_image=src[None].cuda())
    assert maxabs(dm["deformation"].cpu(), g["deformation"]) < 2e-5
    assert maxabs(dm["occlusion_map"].cpu(), g["occlusion_map"]) < 2e-5
    assert maxabs(dm["driving_kp_heatmap"].cpu(), g["driving_kp_heatmap"]) < 1e-5
    assert maxabs(dm["mask"].cpu()[:, :, ::4, ::4], g["mask"]) < 2e-5
    assert maxabs(dm["sparse_deformed"].cpu()[:, :, :, ::4, ::4], g["sparse_deformed"]) < 2e-5
    for k in ("sparse_motion", "kp_heatmap", "source", "kp_driving", "kp_source"):
        assert k in dm


def _netg_inputs(g, sl):
    heat = g["driving_kp_heatmap"]
    return {"deformation": torch.from_numpy(g["deformation"][sl]).cuda(),
            "occlusion_map": torch.from_numpy(g["occlusion_map"][sl]).cuda(),
            "driving_kp_heatmap": torch.from_numpy(heat[sl] if heat.shape[0] > 1 else heat).cuda()}


def test_netg_all_stages_vs_reference(nets):
    net_g, _ = nets
    src, _ = clip()
    g = golden("netg.npz")
    o = net_g(src[None].cuda(), _netg_inputs(golden("dense_motion.npz"), slice(0, 1)), w=1, inference=True)
    for i in range(4):
        assert maxabs(o["out_occ"][i].cpu(), g[f"out_occ_{i}"]) < 1e-4, i
        assert maxabs(o["res_deform_list"][i].cpu(), g[f"res_deform_{i}"]) < 1e-4, i
    for i in range(5):
        assert maxabs(o["deformation_list"][i].cpu(), g[f"deformation_{i}"]) < 1e-4, i
    assert maxabs(o["lq_feat"].cpu(), g["lq_feat"]) < 1e-3
    for key in ("deform_feat_list", "app_comp_list", "app_before_comp_list"):
        for i in range(4):
            assert maxabs(o[key][i].cpu()[:, ::8, ::4, ::4], g[f"{key}_{i}"]) < 1e-3, (key, i)
    assert tuple(o["out"].shape) == (1, 3, 256, 256)
    assert maxabs(o["out"].cpu(), g["out"]) < 1e-3


def test_netg_out_of_frame_flow_and_padding_mask(nets):
    net_g, me = nets
    src, _ = clip()
    g = golden("synthkp.npz")
    kps, kpd = synth_keypoints(2, seed=7)
    dm = me.estimate_motion_w_kp(kp_source=_cuda(kps), kp_driving=_cuda(kpd), source_image=src[None].cuda())
    assert maxabs(dm["deformation"].cpu(), g["deformation"]) < 2e-5
    assert maxabs(dm["occlusion_map"].cpu(), g["occlusion_map"]) < 2e-5
    one = {"deformation": dm["deformation"][1:2].contiguous(), "occlusion_map": dm["occlusion_map"][1:2].contiguous(),
           "driving_kp_heatmap": dm["driving_kp_heatmap"][1:2].contiguous()}
    o = net_g(src[None].cuda(), one, w=1, inference=True)
    assert maxabs(o["deformation_list"][4].cpu(), g["deformation_4"]) < 1e-4
    assert maxabs(o["out_occ"][3].cpu(), g["out_occ_3"]) < 1e-4
    assert maxabs(o["lq_feat"].cpu(), g["lq_feat"]) < 1e-3
    assert maxabs(o["out"].cpu(), g["out"]) < 1e-3


def test_batching_is_exact_up_to_rounding(nets):
    """B frames in one launch == B single-frame calls (no cross-sample coupling in eval)."""
    net_g, _ = nets
    src, _ = clip()
    g = golden("dense_motion.npz")
    both = net_g(src[None].cuda(), _netg_inputs(g, slice(0, 2)), w=1, inference=True)["out"].cpu()
    for i in range(2):
        one = net_g(src[None].cuda(), _netg_inputs(g, slice(i, i + 1)), w=1, inference=True)["out"].cpu()
        assert maxabs(both[i:i + 1], one) < 2e-4, i


def test_config1_eight_frame_clip_vs_reference(nets):
    """BASELINE.json configs[0] on the HIP path: demo.make_animation semantics, uint8 frames."""
    from synergize_motion_appearance_amd.driver import make_animation, animate_batched
    net_g, me = nets
    src, drv = clip()
    g = golden("e2e.npz")
    preds, _ = make_animation(src, list(drv), net_g, me, relative=False, adapt_movement_scale=False, batch=4)
    got = np.stack(preds)
    ref = g["frames_r0a0"]
    assert got.shape == ref.shape and got.dtype == np.uint8
    assert np.abs(got.astype(int) - ref.astype(int)).max() <= 1          # 1e-3 * 127.5 < 1 LSB
    assert (got != ref).mean() < 5e-3
    preds, _ = make_animation(src, list(drv[:4]), net_g, me, relative=True, adapt_movement_scale=True, batch=3)
    assert np.abs(np.stack(preds).astype(int) - g["frames_r1a1"].astype(int)).max() <= 1
    fl = animate_batched(src.cuda(), drv[7:8].cuda(), net_g, me, relative=False, adapt_movement_scale=False, want="float")
    assert maxabs(fl.cpu(), g["out_f7"]) < 1e-3


def test_demo_loop_drop_in(nets):
    """the reference's own per-frame call sequence (demo.py:114-131) runs unchanged on our modules."""
    from basicsr.utils import tensor2img
    net_g, me = nets
    src, drv = clip()
    g = golden("e2e.npz")
    source_img = src.unsqueeze(0).cuda()
    kp_source = me.estimate_kp(source_img)
    for t in range(2):
        kp_driving = me.estimate_kp(drv[t:t + 1].cuda())
        dense_motion = me.estimate_motion_w_kp(kp_source=kp_source, kp_driving=kp_driving, source_image=source_img)
        out_dict = net_g(source_img, dense_motion, w=1, inference=True)
        img = tensor2img([out_dict['out'].detach().cpu()], rgb2bgr=False, min_max=(-1, 1))
        assert np.abs(img.astype(int) - g["frames_r0a0"][t].astype(int)).max() <= 1


def test_full_size_properties(nets):
    """size-independent properties at BASELINE config 2 batch sizes (oracle too slow there):
    identity flow + occlusion 1 => warp is the identity; frames independent of batch position."""
    from synergize_motion_appearance_amd import ops
    B = 16
    feat = torch.randn(1, 256, 256, 64, device="cuda")
    ident = O.make_coordinate_grid(64, 64, torch.float32)[None].repeat(B, 1, 1, 1).cuda()
    w = ops.warp(feat, ident, torch.ones(B, 64, 64, device="cuda"))
    # white-noise features: ~1e-7 rounding of the resized identity flow x 127.5 px x O(1)/px gradient
    assert maxabs(w.cpu(), feat.expand(B, -1, -1, -1).cpu()) < 5e-4
    z = ops.warp(feat, ident + 5.0)                                      # everything out of frame -> zeros
    assert float(z.abs().max()) == 0.0


def test_model_test_side_entries_vs_reference(nets):
    """encode_driving / generator(lq_feat) as called by the reference's model.test()
    (models/appmotioncomp_model.py:450-454)."""
    net_g, _ = nets
    _, drv = clip()
    g, gn = golden("aux_entries.npz"), golden("netg.npz")
    ed = net_g.encode_driving(drv[2:3].cuda())
    assert sorted(ed) == ["128", "256", "32", "64"]
    for k, v in ed.items():
        assert maxabs(v.cpu()[:, ::8, ::4, ::4], g[f"enc_{k}"]) < 5e-4, k
    rec = net_g.generator(torch.from_numpy(gn["lq_feat"]).cuda())
    assert maxabs(rec.cpu(), g["lq_recon"]) < 1e-3
    assert len(net_g.state_dict()) == 472                      # the callable node still owns its parameters


def test_anchor_animation_equals_forward_backward_splice(nets):
    """generate_video_image semantics (models/appmotioncomp_model.py:675-683): backward[::-1] + forward[1:]
    from the anchor frame == one batched pass with kp_driving_initial taken at the anchor."""
    from synergize_motion_appearance_amd.driver import animate_batched
    net_g, me = nets
    src, drv = clip()
    d, a = drv.cuda(), 3
    fwd = animate_batched(src.cuda(), d[a:], net_g, me, True, True, batch=5)
    bwd = animate_batched(src.cuda(), d[:a + 1].flip(0), net_g, me, True, True, batch=5)
    splice = torch.cat([bwd.flip(0), fwd[1:]])
    one = animate_batched(src.cuda(), d, net_g, me, True, True, batch=4, anchor_idx=a)
    assert splice.shape == one.shape
    assert int((splice.int() - one.int()).abs().max()) <= 1


@pytest.mark.parametrize("B", [60, 300])
def test_bench_batch_of_60_equals_single_frame_runs(nets, B):
    """The benchmark's own configurations (bench.py: B = 300 frames per launch sequence, the whole clip; B = 60 in rounds 1-2 -- the wide
    Winograd blocks, GEMM tiles and the row-chunk warp kernel are all selected by launch size): 3 sampled frames of a B-frame batch against
    their B = 1 runs, fp32 <= 2e-4 (different tile shapes change the accumulation order, nothing else), uint8 <= 1 LSB."""
    from synergize_motion_appearance_amd import driver
    from synergize_motion_appearance_amd.synth import synth_clip
    net_g, me = nets
    src, drv = synth_clip(B, seed=123)
    src, drv = src.cuda(), drv.cuda()
    st = driver.encode_source_state(net_g, me, src, drv[0:1], True)
    from synergize_motion_appearance_amd import ops
    with ops.profile() as rec:
        u8, fl = driver.render_frames(st, drv, net_g, me, True, True, batch=B, want="both")
    assert u8.shape == (B, 256, 256, 3) and fl.shape == (B, 3, 256, 256)
    # round 6: at these batch sizes the big 3x3 launches run on the split-bf16 Winograd kernel (the B = 1 runs below on the fp32-MFMA one): the comparison
    # that follows is between the two arithmetic paths
    n_bf3 = sum(1 for r in rec.rows if r[0] == "gemm_conv" and (r[1] or {}).get("bf3") in (4, 6))      # 6: bf16x6, 4: f16x3 (the default since the end of round 6)
    assert ops.WINO_BF3 == 6 and n_bf3 >= 60, n_bf3
    for i in (0, B // 2 + 1, B - 1):
        u1, f1 = driver.render_frames(st, drv[i:i + 1], net_g, me, True, True, batch=1, want="both")
        assert maxabs(fl[i:i + 1].cpu(), f1.cpu()) < 2e-4, i
        assert int((u8[i].int() - u1[0].int()).abs().max()) <= 1, i
    # and the state built on this rank equals a packed / unpacked (broadcast-shaped) copy of itself
    st2 = driver.unpack_source_state(driver.pack_source_state(st.cache, st.src64, st.kp_source, st.kp_initial, st.scale))
    u2 = driver.render_frames(st2, drv[7:9], net_g, me, True, True, batch=2)
    assert torch.equal(u2, driver.render_frames(st, drv[7:9], net_g, me, True, True, batch=2))


def test_source_cache_is_keyed_on_content_not_on_the_pointer(nets):
    """refilling a preallocated source buffer through a raw-pointer writer (this package's own kernels do not
    bump tensor._version) must re-encode; an equal image at another address must hit the cache; and the key must be
    computable under torch.inference_mode()."""
    from synergize_motion_appearance_amd import ops
    from synergize_motion_appearance_amd.synth import synth_clip
    net_g, me = nets
    a, _ = synth_clip(1, seed=5)
    b, _ = synth_clip(1, seed=6)
    buf = a[None].cuda().contiguous()
    c1 = net_g.encode_source(buf)
    assert net_g.encode_source(a[None].cuda()) is c1                        # same content, other address: hit
    v = buf._version
    ops.copy_slice(b[None].cuda().view(1, -1, 1), buf.view(1, -1, 1))       # raw-pointer rewrite of the SAME buffer
    assert buf._version == v and maxabs(buf.cpu(), b[None]) == 0.0
    c2 = net_g.encode_source(buf)
    assert c2 is not c1 and maxabs(c2.feats[32].cpu(), c1.feats[32].cpu()) > 1e-3
    with torch.inference_mode():
        x = b[None].cuda()
        assert net_g.encode_source(x) is c2
    s1 = me._source64(buf)
    ops.copy_slice(a[None].cuda().view(1, -1, 1), buf.view(1, -1, 1))
    assert me._source64(buf) is not s1
    net_g.invalidate_source_cache()
    assert net_g.encode_source(buf) is not c1


def test_frame_pipeline_host_to_host_equals_device_resident_run(nets):
    """row N3: uint8 host frames -> FramePipeline (pinned H2D, device normalise, render, D2H on three streams) must equal
    render_frames on the same frames converted with the reference's own host arithmetic (demo.py:180-185: x/255 then
    (x - 0.5)/0.5 in fp32); ragged last batch; the ingest kernel itself bit-exact vs that arithmetic; and its resize branch
    (cv2.INTER_LINEAR geometry, uint8-rounded) within 1 LSB of a float half-pixel bilinear."""
    import torch.nn.functional as F
    from synergize_motion_appearance_amd import driver, ops
    from synergize_motion_appearance_amd.synth import synth_clip
    net_g, me = nets
    src, drv = synth_clip(11, seed=9)
    u8 = ops.to_uint8(drv.cuda().permute(0, 2, 3, 1).contiguous(), -1.0, 1.0).cpu()          # [11,256,256,3] uint8 "decoded video"
    ref_in = ((u8.float() / 255.0) - 0.5) / 0.5                                               # the reference's host-side arithmetic
    x = ops.frames_u8_to_nchw(u8.cuda())
    assert torch.equal(x.cpu(), ref_in.permute(0, 3, 1, 2))
    assert torch.equal(ops.frames_u8_to_nchw(u8.cuda(), swap_rb=True).cpu(), ref_in.flip(-1).permute(0, 3, 1, 2))
    st = driver.encode_source_state(net_g, me, src.cuda(), x[0:1], True)
    want = driver.render_frames(st, x, net_g, me, True, True, batch=4)
    pipe = driver.FramePipeline(net_g, me, batch=4)
    got = pipe.run(st, u8)
    assert got.dtype == torch.uint8 and not got.is_cuda and torch.equal(got, want.cpu())
    again = torch.cat([c.clone() for _, c in pipe.stream(st, u8.numpy())])                    # numpy in, chunks out, buffers recycled
    assert torch.equal(again, got)
    # the full batches above went through the captured hipGraph (use_graph default), the ragged last one eagerly: bit-identical either way,
    # also without the graph, for a SECOND source through the same pipeline (its packed state copied into the static buffer) and back
    assert pipe._graph is not None
    assert torch.equal(driver.FramePipeline(net_g, me, batch=4, use_graph=False).run(st, u8), got)
    src2, _ = synth_clip(1, seed=10)
    st2 = driver.encode_source_state(net_g, me, src2.cuda(), x[0:1], True)
    want2 = driver.render_frames(st2, x, net_g, me, True, True, batch=4)
    assert torch.equal(pipe.run(st2, u8), want2.cpu()) and not torch.equal(want2, want)
    assert torch.equal(pipe.run(st, u8), got)
    # resize branch: 300x340 frames -> 256x256
    big = (torch.rand((2, 300, 340, 3), generator=torch.Generator().manual_seed(3)) * 255).to(torch.uint8)
    y = ops.frames_u8_to_nchw(big.cuda(), (256, 256))
    ref = F.interpolate(big.permute(0, 3, 1, 2).float(), size=(256, 256), mode="bilinear", align_corners=False)
    back = ((y.cpu() * 0.5 + 0.5) * 255.0)
    assert float((back - back.round()).abs().max()) < 1e-3 and float((back.round() - ref).abs().max()) <= 1.0 + 1e-3


def test_standalone_registered_archs_vs_reference():
    """KPDetector / DenseMotionNetwork / VQGANDiscriminator built by name (the reference registers them too): outputs vs
    the fixture produced by the reference's own standalone modules (tests/golden/make_golden_r2.py)."""
    from basicsr.archs import build_network
    from synergize_motion_appearance_amd.synth import synth_state_dict, synth_clip, synth_input
    from tests.test_host_logic import STANDALONE_OPTS
    nets_ = {}
    for name, opt in STANDALONE_OPTS.items():
        n = build_network(opt)
        n.load_state_dict(synth_state_dict([(k, tuple(v.shape)) for k, v in n.state_dict().items()]), strict=True)
        nets_[name] = n.cuda().eval()
    g = golden("standalone_motion.npz")
    src, drv = synth_clip(3, seed=123)
    kp_s = nets_["KPDetector"](src[None].cuda(), isSource=True)
    kp_d = nets_["KPDetector"](drv.cuda())
    assert maxabs(kp_s["value"].cpu(), g["src_value"]) < 1e-5 and maxabs(kp_s["jacobian"].cpu(), g["src_jacobian"]) < 2e-5
    assert maxabs(kp_d["value"].cpu(), g["drv_value"]) < 1e-5 and maxabs(kp_d["jacobian"].cpu(), g["drv_jacobian"]) < 2e-5
    dm = nets_["DenseMotionNetwork"](source_image=src[None].cuda(), kp_driving=kp_d, kp_source=kp_s)
    assert set(g["keys"].tolist()) <= set(dm.keys())
    assert maxabs(dm["deformation"].cpu(), g["deformation"]) < 2e-5 and maxabs(dm["occlusion_map"].cpu(), g["occlusion_map"]) < 2e-5
    assert maxabs(dm["driving_kp_heatmap"].cpu(), g["driving_kp_heatmap"]) < 1e-5
    assert maxabs(dm["mask"].cpu()[:, :, ::4, ::4], g["mask"]) < 2e-5
    d = nets_["VQGANDiscriminator"](synth_input("disc_in", (2, 3, 64, 64)).cuda())
    ref = golden("discriminator.npz")["out"]
    assert tuple(d.shape) == ref.shape and maxabs(d.cpu(), ref) < 1e-4


def test_batch_of_distinct_sources(nets):
    """configs[2]-style batches: every sample has its OWN source image (x batch == dense_motion batch);
    equals per-sample calls, and sample 0 equals the reference fixture."""
    from synergize_motion_appearance_amd.synth import synth_clip
    net_g, me = nets
    src0, drv = clip()
    src1, _ = synth_clip(1, seed=77)
    srcs = torch.stack([src0, src1]).cuda()
    kp_s = me.estimate_kp(srcs)
    kp_d = me.estimate_kp(drv[[2, 5]].cuda())
    dm = me.estimate_motion_w_kp(kp_source=kp_s, kp_driving=kp_d, source_image=srcs)
    both = net_g(srcs, dm, w=1, inference=True)["out"].cpu()
    for i in range(2):
        kps_i = {k: v[i:i + 1].contiguous() for k, v in kp_s.items()}
        kpd_i = {k: v[i:i + 1].contiguous() for k, v in kp_d.items()}
        dmi = me.estimate_motion_w_kp(kp_source=kps_i, kp_driving=kpd_i, source_image=srcs[i:i + 1].contiguous())
        one = net_g(srcs[i:i + 1].contiguous(), dmi, w=1, inference=True)["out"].cpu()
        assert maxabs(both[i:i + 1], one) < 3e-4, i
    assert maxabs(both[0:1], golden("netg.npz")["out"]) < 1e-3


@pytest.mark.gpu
def test_graft_entry_build_then_smoke_in_a_fresh_process():
    """build() followed by smoke() in ONE fresh interpreter (the driver's order): libsmx must bind
    to the HIP runtime torch brings, whichever of the two is touched first."""
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(repo, "__graft_entry__.py"), "--smoke"], capture_output=True,
                       text=True, timeout=600, cwd=repo)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "smoke: one 256x256 frame" in r.stdout


def test_demo_entry_from_png_folder_end_to_end(nets, tmp_path):
    """`basicsr/demo.py` with the reference's command line: checkpoints named in the yml, a PNG source, a folder of PNG driving
    frames, --relative --adapt_scale --best_frame: the frames it writes are exactly what `animate_batched` renders from the
    same uint8 inputs (anchor semantics = the reference's backward/forward splice, demo.py:205-216), and --visual_video
    holds [source | driving | result] panels."""
    import importlib.util
    from synergize_motion_appearance_amd import ops
    from synergize_motion_appearance_amd.driver import animate_batched
    from synergize_motion_appearance_amd.png import encode_png, decode_png
    net_g, me = nets
    src, drv = clip()
    to_u8 = lambda t: ((t.clamp(-1, 1) + 1) * 127.5).round().to(torch.uint8).permute(0, 2, 3, 1).contiguous().numpy()
    s8, d8 = to_u8(src[None])[0], to_u8(drv)
    (tmp_path / "source.png").write_bytes(encode_png(s8))
    os.makedirs(tmp_path / "drv")
    for i, f in enumerate(d8):
        (tmp_path / "drv" / f"{i:03d}.png").write_bytes(encode_png(f))
    torch.save({"params_ema": {"module." + k: v.cpu() for k, v in net_g.state_dict().items()}}, tmp_path / "g.pth")
    torch.save({"params": {k: v.cpu() for k, v in me.state_dict().items()}}, tmp_path / "m.pth")
    cfg = yaml.safe_load(open(os.path.join(REPO, "options/test.yml")))
    cfg["path"] = {"pretrain_network_g": str(tmp_path / "g.pth"), "param_key_g": "params_ema", "strict_load_g": True,
                   "pretrain_network_motion_estimator": str(tmp_path / "m.pth"), "strict_load_motion_estimator": True}
    yaml.safe_dump(cfg, open(tmp_path / "demo.yml", "w"))
    spec = importlib.util.spec_from_file_location("smx_demo", os.path.join(REPO, "basicsr", "demo.py"))
    demo = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(demo)
    out = demo.main(["--config", str(tmp_path / "demo.yml"), "--source_image", str(tmp_path / "source.png"), "--driving_video", str(tmp_path / "drv"),
                     "--result_video", str(tmp_path / "res.mp4"), "--visual_video", str(tmp_path / "vis.mp4"), "--relative", "--adapt_scale",
                     "--best_frame", "3", "--batch", "4"])
    assert out.shape == (drv.shape[0], 256, 256, 3) and out.dtype == np.uint8
    norm = lambda a: ops.frames_u8_to_nchw(torch.from_numpy(a).cuda(), (256, 256))
    ref = animate_batched(norm(s8[None])[0], norm(d8), net_g, me, True, True, batch=4, anchor_idx=3).cpu().numpy()
    assert np.array_equal(out, ref)                                # same kernels at the same batch size: identical bytes
    written = sorted(os.listdir(str(tmp_path / "res.mp4") + ".frames")) if os.path.isdir(str(tmp_path / "res.mp4") + ".frames") else None
    if written is not None:                                        # no imageio here: PNG frames beside the requested name
        assert len(written) == drv.shape[0]
        assert np.array_equal(decode_png(open(os.path.join(str(tmp_path / "res.mp4") + ".frames", written[2]), "rb").read()), out[2])
        vis = decode_png(open(os.path.join(str(tmp_path / "vis.mp4") + ".frames", written[2]), "rb").read())
        assert vis.shape == (256, 768, 3) and np.array_equal(vis[:, 512:], out[2]) and np.array_equal(vis[:, 256:512], d8[2]) and np.array_equal(vis[:, :256], s8)


def test_demo_entry_streams_a_png_folder_to_a_png_folder(nets, tmp_path):
    """the same entry without --visual_video: PNG folder in, PNG folder out through the streaming path (frames decoded on the codec thread pool straight into the
    pinned staging buffers, rendered, encoded on the pool): the files it writes are exactly what `animate_batched` renders from the same uint8 inputs."""
    import importlib.util
    from synergize_motion_appearance_amd import ops
    from synergize_motion_appearance_amd.driver import animate_batched
    from synergize_motion_appearance_amd.png import encode_png, decode_png
    net_g, me = nets
    src, drv = clip()
    to_u8 = lambda t: ((t.clamp(-1, 1) + 1) * 127.5).round().to(torch.uint8).permute(0, 2, 3, 1).contiguous().numpy()
    s8, d8 = to_u8(src[None])[0], to_u8(drv)
    (tmp_path / "source.png").write_bytes(encode_png(s8))
    os.makedirs(tmp_path / "drv")
    for i, f in enumerate(d8):
        (tmp_path / "drv" / f"{i:03d}.png").write_bytes(encode_png(f))
    cfg = yaml.safe_load(open(os.path.join(REPO, "options/test.yml")))
    yaml.safe_dump(cfg, open(tmp_path / "demo.yml", "w"))
    spec = importlib.util.spec_from_file_location("smx_demo2", os.path.join(REPO, "basicsr", "demo.py"))
    demo = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(demo)
    n = demo.animate_folder(net_g, me, s8, str(tmp_path / "drv"), str(tmp_path / "out"), True, True, anchor=2, batch=3)
    assert n == drv.shape[0]
    norm = lambda a: ops.frames_u8_to_nchw(torch.from_numpy(a).cuda(), (256, 256))
    ref = animate_batched(norm(s8[None])[0], norm(d8), net_g, me, True, True, batch=3, anchor_idx=2).cpu().numpy()
    names = sorted(os.listdir(tmp_path / "out"))
    assert names == [f"{i:06d}.png" for i in range(n)]
    for i, nm in enumerate(names):
        assert np.array_equal(decode_png(open(tmp_path / "out" / nm, "rb").read()), ref[i]), i


def test_checkpoint_like_statistics_end_to_end():
    """parity under the statistics a trained checkpoint has (synth style "checkpoint"): convolutions in front of
    GroupNorm with a large common bias (|group mean| 8-11 x, std ~1 at the first normalisations: where fp32 E[x^2] - mean^2 would
    cancel and where the fp32 Winograd transforms see large common-mode inputs), BatchNorm running statistics far from (0,1), and the
    reference's own U(-1/K, 1/K) codebooks, i.e. quantiser decisions with tiny margins.  Fixture: tests/golden/stress.npz from the
    reference (make_golden_r3.py stress).  fp32 pixels <= 1e-3, uint8 <= 1 LSB, indices tie-aware against the fp64 margins."""
    assert torch.cuda.is_available(), "needs an MI355X"
    import yaml
    from basicsr.archs import build_network
    from synergize_motion_appearance_amd.synth import synth_state_dict, synth_clip
    from tests.util import manifest
    g = golden("stress.npz")
    cfg = yaml.safe_load(open(os.path.join(REPO, "options/test.yml")))
    net_g, me = build_network(cfg["network_g"]), build_network(cfg["network_motion_estimator"])
    net_g.load_state_dict(synth_state_dict([(k, tuple(s)) for k, s in manifest()["network_g"]], style="checkpoint"), strict=True)
    me.load_state_dict(synth_state_dict([(k, tuple(s)) for k, s in manifest()["network_motion_estimator"]], style="checkpoint"), strict=True)
    net_g, me = net_g.cuda().eval(), me.cuda().eval()
    src, drv = synth_clip(8, seed=123)
    idx = g["frames"].tolist()
    s = src[None].cuda()
    kp_s, kp_d = me.estimate_kp(s), me.estimate_kp(drv[idx].cuda())
    assert maxabs(kp_d["value"].cpu(), g["kp_value"]) < 1e-4 and maxabs(kp_d["jacobian"].cpu(), g["kp_jacobian"]) < 5e-4
    dm = me.estimate_motion_w_kp(kp_source=kp_s, kp_driving=kp_d, source_image=s)
    assert maxabs(dm["deformation"].cpu(), g["deformation"]) < 1e-4 and maxabs(dm["occlusion_map"].cpu(), g["occlusion_map"]) < 1e-4
    o = net_g(s, dm, w=1, inference=True)
    assert maxabs(o["deformation_list"][-1].cpu(), g["final_flow"]) < 2e-4
    err = maxabs(o["out"].cpu()[:, :, ::2, ::2], g["out"])
    assert err < 1e-3, err
    u8 = (torch.from_numpy(g["out"]).clamp(-1, 1) + 1) * 127.5
    mine = (o["out"].cpu()[:, :, ::2, ::2].clamp(-1, 1) + 1) * 127.5
    assert int((mine.round() - u8.round()).abs().max()) <= 1
    # the 8 live quantiser calls of the training branch under near-tie codebooks
    dm1 = {k: (v[:1] if torch.is_tensor(v) else v) for k, v in dm.items() if k not in ("kp_driving", "kp_source")}
    ot = net_g(s, dm1, w=1, inference=False, gt=drv[2:3].cuda())
    assert maxabs(ot["out"].cpu()[:, :, ::2, ::2], g["out_train"]) < 1e-3
    mine_idx = {}
    for k, st in zip((256, 512, 768, 1024), ot["_vq_stats_motion"]):
        mine_idx[f"motion:{k}"] = st["min_encoding_indices"].reshape(-1).cpu().numpy()
    for k, st in zip((256, 512, 768, 1024), ot["_vq_stats_app"]):
        mine_idx[f"app:{k}"] = st["min_encoding_indices"].reshape(-1).cpu().numpy()
    total = flipped = 0
    for n, tag in enumerate(str(t) for t in g["vq_order"]):
        ref, margin, got = g[f"vq{n}_indices"], g[f"vq{n}_margin"], mine_idx[tag]
        # distances are sums of D products of O(1/K) codebook entries with O(1) features: fp32 noise ~ 1e-7 * |d|; a decision whose fp64
        # margin is above 2e-6 must agree, below it either of the two nearest codes is a correct answer of the fp32 expression
        safe = margin > 2e-6
        assert (got[safe] == ref[safe]).all(), (tag, int((got[safe] != ref[safe]).sum()), float(margin[safe][got[safe] != ref[safe]].max()))
        total, flipped = total + got.size, flipped + int((got != ref).sum())
    assert flipped <= 0.02 * total, (flipped, total)
