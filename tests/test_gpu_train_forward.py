"""SURVEY row N2, slice 1: the FORWARD of the training branch, `AppMotionCompFormer.forward(x, dense_motion, w, inference=False,
gt=driving)` -- the branch in which VectorQuantizer.forward is live (8 calls per step) -- on the HIP path, against a fixture
produced by the reference's own forward (tests/golden/make_golden_r2.py train_forward; quantizer calls recorded by a spy).
fp32 bars as in the inference tests (<= 1e-3 on pixels); codebook indices bit-exact wherever the reference's own decision
margin (second-best minus best distance, fp64) is above the fp32 noise of the distances -- near-ties are not index-checkable
 and must then still pick one of the two tied codes."""
import os

import numpy as np
import pytest
import torch
import yaml

from tests.util import golden, weights, clip, maxabs, HERE

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(HERE)


@pytest.fixture(scope="module")
def net_g():
    assert torch.cuda.is_available(), "needs an MI355X"
    from basicsr.archs import build_network
    cfg = yaml.safe_load(open(os.path.join(REPO, "options/test.yml")))
    n = build_network(cfg["network_g"])
    n.load_state_dict(weights("network_g"), strict=True)
    return n.eval().cuda()


def test_training_branch_forward_vs_reference(net_g):
    g = golden("train_forward.npz")
    src, drv = clip()
    dm = {"deformation": torch.from_numpy(g["deformation"]).cuda(), "occlusion_map": torch.from_numpy(g["occlusion_map"]).cuda(),
          "driving_kp_heatmap": torch.from_numpy(g["driving_kp_heatmap"]).cuda()}
    o = net_g(src[None].cuda(), dm, w=1, inference=False, gt=drv[2:3].cuda())
    # same image as the inference branch, plus the un-fused second decoder pass
    assert maxabs(o["out"].cpu(), g["out"]) < 1e-3 and maxabs(o["out_lr"][0].cpu(), g["out_lr"]) < 1e-3
    ref_inf = net_g(src[None].cuda(), dm, w=1, inference=True)
    assert torch.equal(o["out"], ref_inf["out"])                       # the training outputs ride along, the image path is untouched
    assert "out_lr" not in ref_inf and "motion_recon_list" not in ref_inf
    # to_motion(quantize_motion(m_feat)) per scale, codebook losses
    assert len(o["motion_recon_list"]) == 4 and len(o["codebook_loss_motion_list"]) == 4
    for i in range(4):
        assert tuple(o["motion_recon_list"][i].shape) == (1, 64, 64, 2)
        assert maxabs(o["motion_recon_list"][i].cpu(), g[f"motion_recon_{i}"]) < 1e-4, i
        assert abs(float(o["codebook_loss_motion_list"][i]) - g["codebook_loss_motion"][i]) < 1e-4 * max(1.0, abs(g["codebook_loss_motion"][i])), i
    # app_codebook_loss(gt)
    assert len(o["app_recon_list"]) == 4 and len(o["codebook_loss_app_list"]) == 4
    for i, row in enumerate(o["app_recon_list"]):
        app_recon, app_orig, quant_app, app_feat, feat_com = row
        assert maxabs(feat_com.cpu()[:, ::8, ::4, ::4], g[f"feat_com_{i}"]) < 5e-4, i
        assert maxabs(app_feat.cpu()[:, ::8], g[f"app_feat_{i}"]) < 5e-4, i
        assert maxabs(app_orig.cpu()[:, ::8, ::4, ::4], g[f"app_feat_original_{i}"]) < 1e-3, i
        assert abs(float(o["codebook_loss_app_list"][i]) - g["codebook_loss_app"][i]) < 2e-4 * max(1.0, abs(g["codebook_loss_app"][i])), i
    # the 8 live VectorQuantizer calls: indices vs the reference's, tie-aware
    order = [str(s) for s in g["vq_order"]]
    mine = {}
    for k, st in zip((256, 512, 768, 1024), o["_vq_stats_motion"]):
        mine[f"motion:{k}"] = st["min_encoding_indices"].reshape(-1).cpu().numpy()
    for k, st in zip((256, 512, 768, 1024), o["_vq_stats_app"]):
        mine[f"app:{k}"] = st["min_encoding_indices"].reshape(-1).cpu().numpy()
    assert sorted(order) == sorted(mine)
    total, flipped = 0, 0
    for n, tag in enumerate(order):
        ref_idx, margin = g[f"vq{n}_indices"], g[f"vq{n}_margin"]
        got = mine[tag]
        assert got.shape == ref_idx.shape and got.dtype == np.int64
        safe = margin > 1e-3 * (1.0 if tag.startswith("motion") else 16.0)      # distances are O(32) / O(256) sums of fp32 products
        assert (got[safe] == ref_idx[safe]).all(), (tag, int((got[safe] != ref_idx[safe]).sum()))
        total += got.size
        flipped += int((got != ref_idx).sum())
    assert flipped <= 0.01 * total, (flipped, total)                       # near-ties only
    # quantised features: z_q of the app calls equals the codebook rows the indices name
    cb = weights("network_g")["quantize_app.embedding.weight"]
    for i, k in enumerate((256, 512, 768, 1024)):
        zq = o["app_recon_list"][i][2].cpu()                                # [1,256,32,32]
        idx = torch.from_numpy(mine[f"app:{k}"])
        z = o["app_recon_list"][i][3].cpu().permute(0, 2, 3, 1).reshape(-1, 256)
        e = cb[idx]
        assert torch.equal(zq.permute(0, 2, 3, 1).reshape(-1, 256), z + (e - z))   # straight-through form, bit-exact
        assert int(idx.max()) < k
