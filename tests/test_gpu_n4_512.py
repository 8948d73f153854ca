"""N4 / BASELINE configs[3]: the 512x512 variant end to end on the MI355X against the CPU oracle.

NEW SEMANTICS, PARITY UNPINNED BY THE REFERENCE: the reference raises at img_size 512 (archs/appmotioncodebook_arch.py:211-216, 573),
so there is no fixture to generate.  The definition (DESIGN.md "N4": the test.yml network with every grid doubled, module names of
the 256 layout, 64x64 token grid, 128x128 flow grid) lives in oracle/reenact_oracle.py, whose functions take their grid sizes from
the inputs and are fixture-checked against the reference at 256; these tests hold the HIP path to that oracle at 512 with the
north-star bar (fp32 pixels <= 1e-3, uint8 <= 1 LSB) and to the size-independent properties of the path."""
import os

import pytest
import torch
import yaml

from oracle import reenact_oracle as O
from tests.util import maxabs, HERE
from synergize_motion_appearance_amd.synth import synth_clip, synth_state_dict

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(HERE)


@pytest.fixture(scope="module")
def nets512():
    assert torch.cuda.is_available(), "needs an MI355X"
    from basicsr.archs import build_network
    cfg = yaml.safe_load(open(os.path.join(REPO, "options/test_512.yml")))
    net_g, me = build_network(cfg["network_g"]), build_network(cfg["network_motion_estimator"])
    Pg = synth_state_dict([(k, tuple(v.shape)) for k, v in net_g.state_dict().items()])
    Pm = synth_state_dict([(k, tuple(v.shape)) for k, v in me.state_dict().items()])
    net_g.load_state_dict(Pg, strict=True)
    me.load_state_dict(Pm, strict=True)
    return net_g.eval().cuda(), me.eval().cuda(), Pg, Pm


def test_state_dict_is_the_256_layout_but_for_the_position_rows(nets512):
    from tests.util import manifest
    net_g, me, _, _ = nets512
    ref = {k: tuple(s) for k, s in manifest()["network_g"]}
    mine = {k: tuple(v.shape) for k, v in net_g.state_dict().items()}
    assert list(mine) == list(ref)
    diff = {k for k in ref if ref[k] != mine[k]}
    assert diff == {"position_emb_app", "position_emb_motion"} and mine["position_emb_app"] == (4096, 256) and mine["position_emb_motion"] == (4096, 32)
    assert {k: tuple(v.shape) for k, v in me.state_dict().items()} == {k: tuple(s) for k, s in manifest()["network_motion_estimator"]}


def test_512_stages_and_pixels_vs_oracle(nets512):
    """keypoints (128x128 hourglass, 122x122 heatmaps), dense motion (128x128 flow / occlusion), the four compensation scales and the
    decoded 512x512 frame against the oracle's definition, two driving frames in one batch."""
    net_g, me, Pg, Pm = nets512
    src, drv = synth_clip(3, seed=77, size=512)
    s, d = src[None], drv[1:3]
    with torch.no_grad():
        kp_s_o, kp_d_o = O.kp_detector(Pm, s), O.kp_detector(Pm, d)
        dm_o = O.dense_motion(Pm, s.expand(2, -1, -1, -1), kp_d_o, {k: v.expand(2, *v.shape[1:]) for k, v in kp_s_o.items()})
        ref = O.netg_forward(Pg, s.expand(2, -1, -1, -1), dm_o)
    assert tuple(dm_o["deformation"].shape) == (2, 128, 128, 2) and tuple(ref["out"].shape) == (2, 3, 512, 512)
    kp_s, kp_d = me.estimate_kp(s.cuda()), me.estimate_kp(d.cuda())
    assert maxabs(kp_d["value"].cpu(), kp_d_o["value"]) < 1e-4 and maxabs(kp_d["jacobian"].cpu(), kp_d_o["jacobian"]) < 5e-4
    dm = me.estimate_motion_w_kp(kp_source=kp_s, kp_driving=kp_d, source_image=s.cuda())
    assert tuple(dm["deformation"].shape) == (2, 128, 128, 2) and tuple(dm["occlusion_map"].shape) == (2, 1, 128, 128)
    assert maxabs(dm["deformation"].cpu(), dm_o["deformation"]) < 1e-4
    assert maxabs(dm["occlusion_map"].cpu(), dm_o["occlusion_map"]) < 1e-4
    o = net_g(s.cuda(), dm, w=1, inference=True)
    assert [tuple(t.shape[1:]) for t in o["app_comp_list"]] == [(256, 64, 64), (128, 128, 128), (128, 256, 256), (64, 512, 512)]
    for i in range(4):
        assert maxabs(o["deformation_list"][i + 1].cpu(), ref["deformation_list"][i + 1]) < 2e-4, i
        assert maxabs(o["out_occ"][i].cpu(), ref["out_occ"][i]) < 2e-4, i
        e = maxabs(o["app_comp_list"][i].cpu(), ref["app_comp_list"][i])
        assert e < 1e-3 * max(1.0, float(ref["app_comp_list"][i].abs().max())), (i, e)
    err = maxabs(o["out"].cpu(), ref["out"])
    assert err < 1e-3, err
    a = ((o["out"].cpu().clamp(-1, 1) + 1) * 127.5).round()
    b = ((ref["out"].clamp(-1, 1) + 1) * 127.5).round()
    assert int((a - b).abs().max()) <= 1


def test_512_pipeline_batching_and_packed_state(nets512):
    """the driver path at 512: the packed frame-invariant state (4x the encoder taps + a 128x128 source) round-trips, frames are
    independent given that state (a batch of 3 == three single-frame runs up to batch-shape rounding), and the host-to-host
    FramePipeline returns the device-resident result."""
    from synergize_motion_appearance_amd import driver, ops
    net_g, me, _, _ = nets512
    src, drv = synth_clip(3, seed=5, size=512)
    st = driver.encode_source_state(net_g, me, src.cuda(), drv[0].cuda(), True)
    flat = driver.pack_source_state(st.cache, st.src64, st.kp_source, st.kp_initial, st.scale)
    assert flat.numel() == driver.cache_numel(torch.float32, 512) == 4 * (7077888 + 12288) + 180 + 1
    st2 = driver.unpack_source_state(flat, torch.float32, 512)
    assert all(torch.equal(st2.cache.feats[k], st.cache.feats[k]) for k in st.cache.feats) and torch.equal(st2.src64, st.src64)
    with pytest.raises(ValueError):
        driver.unpack_source_state(flat)                                   # a 512 state is not a 256 state
    whole = driver.render_frames(st2, drv.cuda(), net_g, me, batch=3, want="float")
    single = torch.cat([driver.render_frames(st2, drv[i:i + 1].cuda(), net_g, me, batch=1, want="float") for i in range(3)])
    # batch shape picks different GEMM / Winograd block shapes (different summation orders): rounding-level, half the 1e-3 pixel bar at most
    assert tuple(whole.shape) == (3, 3, 512, 512) and maxabs(whole.cpu(), single.cpu()) < 5e-4
    u8 = ((drv.permute(0, 2, 3, 1) + 1) * 127.5).round().clamp(0, 255).to(torch.uint8)
    pipe = driver.FramePipeline(net_g, me, batch=2, frame_hw=(512, 512))
    host = pipe.run(st2, u8)
    x = ops.frames_u8_to_nchw(u8.cuda(), (512, 512))
    dev = driver.render_frames(st2, x, net_g, me, batch=2, want="uint8")
    assert tuple(host.shape) == (3, 512, 512, 3) and int((host.int() - dev.cpu().int()).abs().max()) <= 1


def test_512_bf16_storage_within_the_oracles_own_autocast_error(nets512):
    """configs[3] in bf16 (bf16 NHWC storage + bf16 MFMA, as configs[2] at 256): no 1e-3 claim is possible in bf16; the bar is the
    error the 512 DEFINITION itself (the oracle) makes under CPU autocast(bfloat16) on the same inputs, x1.25 -- the rule the 256 path
    is held to with the reference's autocast fixture (tests/test_gpu_bf16.py)."""
    net_g, me, Pg, Pm = nets512
    src, drv = synth_clip(3, seed=77, size=512)
    s, d = src[None], drv[1:2]
    with torch.no_grad():
        kp_s_o, kp_d_o = O.kp_detector(Pm, s), O.kp_detector(Pm, d)
        dm_o = O.dense_motion(Pm, s, kp_d_o, kp_s_o)
        ref = O.netg_forward(Pg, s, dm_o)["out"]
        with torch.autocast("cpu", dtype=torch.bfloat16):
            auto = O.netg_forward(Pg, s, dm_o)["out"].float()
    rmax, rmean = float((auto - ref).abs().max()), float((auto - ref).abs().mean())
    net_g.set_compute_dtype("bf16")
    me.set_compute_dtype("bf16")
    try:
        kp_s, kp_d = me.estimate_kp(s.cuda()), me.estimate_kp(d.cuda())
        dm = me.estimate_motion_w_kp(kp_source=kp_s, kp_driving=kp_d, source_image=s.cuda())
        out = net_g(s.cuda(), dm, w=1, inference=True)["out"].float().cpu()
    finally:
        net_g.set_compute_dtype("f32")
        me.set_compute_dtype("f32")
    emax, emean = float((out - ref).abs().max()), float((out - ref).abs().mean())
    print(f"512 bf16: max {emax:.4f} mean {emean:.5f}   oracle under autocast: max {rmax:.4f} mean {rmean:.5f}")
    assert emax <= 1.25 * rmax and emean <= 1.25 * rmean, (emax, emean, rmax, rmean)
