"""Pin the CPU oracle (oracle/reenact_oracle.py) against fixtures produced by the imported
reference (tests/golden/make_golden.py).  CPU only; tolerances ~ the reference's own fp32
noise (1.8e-5 on `out`)."""
import numpy as np
import torch

from oracle import reenact_oracle as O
from tests.util import golden, weights, clip, maxabs
from synergize_motion_appearance_amd.synth import synth_keypoints, synth_input

torch.set_num_threads(8)


def _kp(i=None):
    g = golden("kp.npz")
    if i is None:
        return {"value": torch.from_numpy(g["src_value"]), "jacobian": torch.from_numpy(g["src_jacobian"])}
    return {"value": torch.from_numpy(g["drv_value"][i]), "jacobian": torch.from_numpy(g["drv_jacobian"][i])}


def test_kp_detector():
    Pm = weights("network_motion_estimator")
    src, drv = clip()
    g = golden("kp.npz")
    with torch.no_grad():
        s = O.kp_detector(Pm, src[None])
        d = O.kp_detector(Pm, drv)
    assert maxabs(s["value"], g["src_value"]) < 2e-6
    assert maxabs(s["jacobian"], g["src_jacobian"]) < 5e-6
    assert maxabs(d["value"], g["drv_value"]) < 2e-6
    assert maxabs(d["jacobian"], g["drv_jacobian"]) < 5e-6


def test_normalize_kp_four_flag_combinations():
    g = golden("normalize_kp.npz")
    for rel in (0, 1):
        for ad in (0, 1):
            r = O.normalize_kp(_kp(), _kp([3]), _kp([0]), bool(ad), bool(rel), bool(rel))
            assert maxabs(r["value"], g[f"value_r{rel}a{ad}"]) < 1e-6
            assert maxabs(r["jacobian"], g[f"jacobian_r{rel}a{ad}"]) < 1e-5


def test_dense_motion():
    Pm = weights("network_motion_estimator")
    src, _ = clip()
    g = golden("dense_motion.npz")
    kps = {k: v.repeat(2, *([1] * (v.dim() - 1))) for k, v in _kp().items()}
    with torch.no_grad():
        dm = O.dense_motion(Pm, src[None].repeat(2, 1, 1, 1), _kp([2, 5]), kps)
    assert maxabs(dm["deformation"], g["deformation"]) < 5e-6
    assert maxabs(dm["occlusion_map"], g["occlusion_map"]) < 5e-6
    assert maxabs(dm["driving_kp_heatmap"], g["driving_kp_heatmap"]) < 1e-6
    assert maxabs(dm["mask"][:, :, ::4, ::4], g["mask"]) < 5e-6
    assert maxabs(dm["sparse_deformed"][:, :, :, ::4, ::4], g["sparse_deformed"]) < 1e-5


def _dm_from(g, sl):
    return {"deformation": torch.from_numpy(g["deformation"][sl]),
            "occlusion_map": torch.from_numpy(g["occlusion_map"][sl]),
            "driving_kp_heatmap": torch.from_numpy(g["driving_kp_heatmap"][sl] if g["driving_kp_heatmap"].shape[0] > 1
                                                   else g["driving_kp_heatmap"])}


def test_netg_all_stages():
    Pg = weights("network_g")
    src, _ = clip()
    g = golden("netg.npz")
    with torch.no_grad():
        o = O.netg_forward(Pg, src[None], _dm_from(golden("dense_motion.npz"), slice(0, 1)))
    for i in range(4):
        assert maxabs(o["out_occ"][i], g[f"out_occ_{i}"]) < 2e-5, i
        assert maxabs(o["res_deform_list"][i], g[f"res_deform_{i}"]) < 2e-5, i
    for i in range(5):
        assert maxabs(o["deformation_list"][i], g[f"deformation_{i}"]) < 2e-5, i
    assert maxabs(o["lq_feat"], g["lq_feat"]) < 2e-4
    for key in ("deform_feat_list", "app_comp_list", "app_before_comp_list"):
        for i in range(4):
            assert maxabs(o[key][i][:, ::8, ::4, ::4], g[f"{key}_{i}"]) < 5e-4, (key, i)
    assert maxabs(o["out"], g["out"]) < 2e-4


def test_netg_out_of_frame_flow_and_padding_mask():
    """synthetic-keypoint mode: ~6% of flow samples outside [-1,1] (zeros padding) and 80
    masked tokens in the layer-0 self-attention key_padding_mask."""
    Pg, Pm = weights("network_g"), weights("network_motion_estimator")
    src, _ = clip()
    g = golden("synthkp.npz")
    kps, kpd = synth_keypoints(2, seed=7)
    with torch.no_grad():
        dm = O.dense_motion(Pm, src[None].repeat(2, 1, 1, 1), kpd, kps)
        assert maxabs(dm["deformation"], g["deformation"]) < 5e-6
        assert maxabs(dm["occlusion_map"], g["occlusion_map"]) < 5e-6
        o = O.netg_forward(Pg, src[None], {k: dm[k][1:2] for k in ("deformation", "occlusion_map", "driving_kp_heatmap")})
    assert maxabs(o["deformation_list"][4], g["deformation_4"]) < 2e-5
    assert maxabs(o["out_occ"][3], g["out_occ_3"]) < 2e-5
    assert maxabs(o["lq_feat"], g["lq_feat"]) < 2e-4
    assert maxabs(o["out"], g["out"]) < 2e-4


def test_make_animation_config1():
    """BASELINE.json configs[0]: 1 source + 8-frame clip through the demo.py loop."""
    Pg, Pm = weights("network_g"), weights("network_motion_estimator")
    src, drv = clip()
    g = golden("e2e.npz")
    with torch.no_grad():
        fr, outs = O.make_animation(Pg, Pm, src, drv[:3], relative=False, adapt_movement_scale=False)
        fr_rel, _ = O.make_animation(Pg, Pm, src, drv[:2], relative=True, adapt_movement_scale=True)
    for t in range(3):
        assert np.abs(fr[t].astype(int) - g["frames_r0a0"][t].astype(int)).max() <= 1
        assert (fr[t] != g["frames_r0a0"][t]).mean() < 1e-3
    for t in range(2):
        assert np.abs(fr_rel[t].astype(int) - g["frames_r1a1"][t].astype(int)).max() <= 1


def test_generate_video_frames_anchor_splice_vs_reference_model_class():
    """the oracle's literal restatement of `generate_video_image`'s anchor splice vs
    the frames the reference's own AppMotionCompModel wrote (tests/golden/make_golden_model.py)."""
    g = golden("model_animate.npz")
    n, anchor, seed = int(g["n_frames"]), int(g["anchor_idx"]), int(g["seed"])
    src, drv = clip(n, seed)
    with torch.no_grad():
        fr = O.generate_video_frames(weights("network_g"), weights("network_motion_estimator"), src, drv, anchor,
                                     relative=True, adapt_movement_scale=True)
    assert len(fr) == n
    for t in range(n):
        d = np.abs(fr[t].astype(int) - g["result_png"][t].astype(int))
        assert d.max() <= 1 and (d > 0).mean() < 1e-3, (t, d.max(), (d > 0).mean())


def test_warp_explicit_matches_aten():
    x = synth_input("warp_feat", (2, 16, 64, 64))
    flow = O.make_coordinate_grid(64, 64, torch.float32)[None].repeat(2, 1, 1, 1) + 0.3 * synth_input("warp_flow", (2, 64, 64, 2))
    occ = torch.sigmoid(synth_input("warp_occ", (2, 1, 64, 64)))
    a = O.occlude_input(O.deform_input(x, flow), occ)
    b = O.warp_explicit(x, flow, occ)
    assert maxabs(a, b) < 1e-5
    x32 = synth_input("warp_feat32", (2, 16, 32, 32))
    assert maxabs(O.deform_input(x32, flow), O.warp_explicit(x32, flow)) < 1e-5


def test_vector_quantizer():
    Pg = weights("network_g")
    g = golden("vq.npz")
    for tag, key, D, scale in (("m256", "quantize_motion", 32, 0.25), ("m1024", "quantize_motion", 32, 1.0),
                               ("a512", "quantize_app", 256, 0.5), ("a1024", "quantize_app", 256, None)):
        z = synth_input(f"vq_{tag}", (2, D, 32, 32))
        r = O.vector_quantizer(z, Pg[f"{key}.embedding.weight"], scale)
        assert np.array_equal(r["indices"].numpy(), g[f"{tag}_indices"]), tag
        assert maxabs(r["loss"], g[f"{tag}_loss"]) < 1e-5
        assert maxabs(r["z_q"][:, :, ::4, ::4], g[f"{tag}_zq_sub"]) == 0.0
        assert maxabs(r["perplexity"], g[f"{tag}_perplexity"]) < 1e-2
        assert maxabs(r["mean_distance"], g[f"{tag}_mean_distance"]) < 1e-3


def test_tensor2img():
    t = synth_input("tensor2img", (3, 64, 64)) * 0.8
    assert np.array_equal(O.tensor2img(t[None]), golden("tensor2img.npz")["img"])


def test_encode_driving_and_generator_entries():
    """side entries used by the reference's model.test() (models/appmotioncomp_model.py:450-454)."""
    Pg = weights("network_g")
    _, drv = clip()
    g, gn = golden("aux_entries.npz"), golden("netg.npz")
    with torch.no_grad():
        ed = O.encode_driving(Pg, drv[2:3])
        rec = O.generator(Pg, torch.from_numpy(gn["lq_feat"]))
    assert sorted(ed) == ["128", "256", "32", "64"]
    for k, v in ed.items():
        assert maxabs(v[:, ::8, ::4, ::4], g[f"enc_{k}"]) < 1e-4, k
    assert maxabs(rec, g["lq_recon"]) < 2e-4
