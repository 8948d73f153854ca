"""Checkpoint key layout (names, shapes) of the two networks as a function of the yml kwargs.

The drop-in contract is a *strict* `load_state_dict` of the reference's checkpoints
(reference `basicsr/demo.py:46-72`, `options/test.yml:72,75`), so parameter and buffer names
and shapes must equal the reference's.  They are generated here from the hyper-parameters
as flat tables (no module classes mirrored) and verified entry-by-entry against
`tests/golden/manifest.json`, which was dumped from the imported reference.
"""

# block kinds of the 19-block encoder / generator for a given ch_mult / attn_resolutions
# (reference `archs/vqgan_arch.py:256-350`)


def encoder_plan(nf, ch_mult, res_blocks, resolution, attn_resolutions):
    """[(kind, cin, cout)] for encoder.blocks.{i}."""
    plan = [("conv", 3, nf)]
    in_mult = (1,) + tuple(ch_mult)
    cur = resolution
    cin = nf
    for i in range(len(ch_mult)):
        cin, cout = nf * in_mult[i], nf * ch_mult[i]
        for _ in range(res_blocks):
            plan.append(("res", cin, cout))
            cin = cout
            if cur in attn_resolutions:
                plan.append(("attn", cin, cin))
        if i != len(ch_mult) - 1:
            plan.append(("down", cin, cin))
            cur //= 2
    plan += [("res", cin, cin), ("attn", cin, cin), ("res", cin, cin), ("gn", cin, cin)]
    return plan, cin


def generator_plan(nf, ch_mult, res_blocks, img_size, attn_resolutions, emb_dim):
    cin = nf * ch_mult[-1]
    cur = img_size // 2 ** (len(ch_mult) - 1)
    plan = [("conv", emb_dim, cin), ("res", cin, cin), ("attn", cin, cin), ("res", cin, cin)]
    for i in reversed(range(len(ch_mult))):
        cout = nf * ch_mult[i]
        for _ in range(res_blocks):
            plan.append(("res", cin, cout))
            cin = cout
            if cur in attn_resolutions:
                plan.append(("attn", cin, cin))
        if i != 0:
            plan.append(("up", cin, cin))
            cur *= 2
    plan += [("gn", cin, cin), ("conv", cin, 3)]
    return plan


def _conv(out, name, cout, cin, k):
    out.append((name + ".weight", (cout, cin, k, k)))
    out.append((name + ".bias", (cout,)))


def _norm(out, name, c):
    out.append((name + ".weight", (c,)))
    out.append((name + ".bias", (c,)))


def _res(out, name, cin, cout):
    _norm(out, name + ".norm1", cin)
    _conv(out, name + ".conv1", cout, cin, 3)
    _norm(out, name + ".norm2", cout)
    _conv(out, name + ".conv2", cout, cout, 3)
    if cin != cout:
        _conv(out, name + ".conv_out", cout, cin, 1)


def _blocks(out, prefix, plan):
    for i, (kind, cin, cout) in enumerate(plan):
        n = f"{prefix}.{i}"
        if kind == "conv":
            _conv(out, n, cout, cin, 3)
        elif kind == "res":
            _res(out, n, cin, cout)
        elif kind == "attn":
            _norm(out, n + ".norm", cin)
            for p in ("q", "k", "v", "proj_out"):
                _conv(out, f"{n}.{p}", cin, cin, 1)
        elif kind in ("down", "up"):
            _conv(out, n + ".conv", cin, cin, 3)
        elif kind == "gn":
            _norm(out, n, cin)


def _transformer(out, name, E):
    for a in ("self_attn", "cross_attn"):
        out.append((f"{name}.{a}.in_proj_weight", (3 * E, E)))
        out.append((f"{name}.{a}.in_proj_bias", (3 * E,)))
        out.append((f"{name}.{a}.out_proj.weight", (E, E)))
        out.append((f"{name}.{a}.out_proj.bias", (E,)))
    _conv(out, name + ".conv1", 2 * E, E, 3)
    _conv(out, name + ".conv2", E, 2 * E, 3)
    for n in ("norm1", "norm2", "norm3"):
        _norm(out, f"{name}.{n}", E)


CHANNELS = {"32": 256, "64": 128, "128": 128, "256": 64}   # appmotioncodebook_arch.py:211-216


def netg_manifest(img_size=256, nf=64, ch_mult=(1, 2, 2, 4), res_blocks=2, attn_resolutions=(32,),
                  codebook_size_motion=1024, embed_dim_motion=32, codebook_size_app=1024, embed_dim_app=256,
                  dim_embd_motion=32, n_layers_motion=2, dim_embd_app=256, n_layers_app=2, split=1, num_kp=15,
                  connect_list=("64", "128", "256"), connect_app_list=("32", "64", "128", "256"), emb_dim=256):
    """[(name, shape)] of AppMotionCompFormer for the options/test.yml flag set
    (with_position_emb, warp_s_d_kp_query, MRFA_motion_enc, multiscale_sft/feature_fusion)."""
    g = img_size // 8            # token grid: 32 in the reference (appmotioncodebook_arch.py:266-267); 64 for the 512 variant (DESIGN "N4")
    out = [("position_emb_app", (g * g, dim_embd_app)), ("position_emb_motion", (g * g, dim_embd_motion))]
    enc, _ = encoder_plan(nf, ch_mult, res_blocks, img_size, attn_resolutions)
    enc = enc + [("conv", enc[-1][1], emb_dim)]
    _blocks(out, "encoder.blocks", enc)
    _blocks(out, "generator.blocks", generator_plan(nf, ch_mult, res_blocks, img_size, attn_resolutions, emb_dim))
    for s in connect_app_list:
        c = CHANNELS[s] // split
        p = int(s) // 32
        if s == "32":
            _conv(out, "app_feat_emb_32", dim_embd_app, c, 1)
            _conv(out, "to_app_feat_32", c, dim_embd_app, 1)
        else:
            out += [(f"app_feat_emb_{s}.1.weight", (dim_embd_app, c * p * p)), (f"app_feat_emb_{s}.1.bias", (dim_embd_app,)),
                    (f"to_app_feat_{s}.0.weight", (c * p * p, dim_embd_app)), (f"to_app_feat_{s}.0.bias", (c * p * p,))]
    out.append(("quantize_app.embedding.weight", (codebook_size_app, embed_dim_app)))
    for s in connect_list:
        c = CHANNELS[s]
        _res(out, f"fuse_convs_dict.{s}.encode_enc", 2 * c, c)
        for br in ("scale", "shift"):
            _conv(out, f"fuse_convs_dict.{s}.{br}.0", c, c, 3)
            _conv(out, f"fuse_convs_dict.{s}.{br}.2", c, c, 3)
    for s in connect_list:
        _conv(out, f"fuse_ms_dict.{s}", CHANNELS[s], CHANNELS[s], 3)
    out.append(("quantize_motion.embedding.weight", (codebook_size_motion, embed_dim_motion)))
    E = dim_embd_motion
    _conv(out, "motion_emb.0", E, 2, 3)
    _conv(out, "motion_emb.1.conv", E, E, 3)
    _res(out, "motion_emb.2", E, E)
    for l in range(n_layers_motion):
        _transformer(out, f"motion_block.{l}", E)
    _conv(out, "to_motion.0.conv", E, E, 3)
    _res(out, "to_motion.1", E, E)
    _norm(out, "to_motion.2", E)
    _conv(out, "to_motion.3", 2, E, 3)
    _conv(out, "BasicMotionEncoder.convc1", 128, E, 1)
    _conv(out, "BasicMotionEncoder.convc2", 96, 128, 3)
    _conv(out, "BasicMotionEncoder.convf1", 128, 2, 7)
    _conv(out, "BasicMotionEncoder.convf2", 64, 128, 3)
    _conv(out, "BasicMotionEncoder.conv", 126, 160, 3)
    ctx = ["32", "64", "128"] + (["256"] if "256" in connect_list else [])
    for i, s in enumerate(ctx):
        _conv(out, f"to_context.{i}", 192, CHANNELS[s], 1)
    _conv(out, "refine.convc1", 128, 192, 3)
    _conv(out, "refine.conv1", 128, 256, 3)
    _conv(out, "refine.conv2", 2, 128, 3)
    _conv(out, "refine.convo1", 128, 256, 3)
    _conv(out, "refine.convo2", 1, 128, 3)
    for l in range(n_layers_app):
        _transformer(out, f"app_block.{l}", dim_embd_app)
    for s in ctx:
        _conv(out, f"warped_source_enc_{s}", E, CHANNELS[s], 1)
    _conv(out, "driving_kp_enc", E, num_kp, 1)
    _conv(out, "motion_query_enc_1", E, 2 * E, 1)
    _conv(out, "motion_query_enc_2", E, 2 * E, 1)
    return out


def hourglass_channels(block_expansion, in_features, num_blocks, max_features):
    """[(cin, cout)] for down blocks and up blocks (utils/motion_estimator_util.py:440-480)."""
    down = [(in_features if i == 0 else min(max_features, block_expansion * 2 ** i),
             min(max_features, block_expansion * 2 ** (i + 1))) for i in range(num_blocks)]
    up = []
    for i in reversed(range(num_blocks)):
        cin = (1 if i == num_blocks - 1 else 2) * min(max_features, block_expansion * 2 ** (i + 1))
        up.append((cin, min(max_features, block_expansion * 2 ** i)))
    return down, up, block_expansion + in_features


def _hourglass(out, name, block_expansion, in_features, num_blocks, max_features):
    down, up, of = hourglass_channels(block_expansion, in_features, num_blocks, max_features)
    for grp, chans in (("encoder.down_blocks", down), ("decoder.up_blocks", up)):
        for i, (cin, cout) in enumerate(chans):
            n = f"{name}.{grp}.{i}"
            _conv(out, n + ".conv", cout, cin, 3)
            _norm(out, n + ".norm", cout)
            out += [(n + ".norm.running_mean", (cout,)), (n + ".norm.running_var", (cout,)),
                    (n + ".norm.num_batches_tracked", ())]
    return of


def motion_estimator_manifest(common_params, dense_motion_params, kp_detector_params):
    """[(name, shape)] of Motion_Estimator_keypoint_aware (kp_detector + dense_motion_network)."""
    num_kp, nc = common_params["num_kp"], common_params["num_channels"]
    out = []
    kp = kp_detector_params
    of = _hourglass(out, "kp_detector.predictor", kp["block_expansion"], nc, kp["num_blocks"], kp["max_features"])
    out += [("kp_detector.kp.weight", (num_kp, of, 7, 7)), ("kp_detector.kp.bias", (num_kp,))]
    if kp.get("estimate_jacobian", False):
        out += [("kp_detector.jacobian.weight", (4 * num_kp, of, 7, 7)), ("kp_detector.jacobian.bias", (4 * num_kp,))]
    if kp.get("scale_factor", 1) != 1:
        out.append(("kp_detector.down.weight", (nc, 1, 13, 13)))
    dm = dense_motion_params
    of = _hourglass(out, "dense_motion_network.hourglass", dm["block_expansion"], (num_kp + 1) * (nc + 1),
                    dm["num_blocks"], dm["max_features"])
    out += [("dense_motion_network.mask.weight", (num_kp + 1, of, 7, 7)), ("dense_motion_network.mask.bias", (num_kp + 1,))]
    if dm.get("scale_factor", 1) != 1:
        out.append(("dense_motion_network.down.weight", (nc, 1, 13, 13)))
    if dm.get("estimate_occlusion_map", False):
        out += [("dense_motion_network.occlusion.weight", (1, of, 7, 7)), ("dense_motion_network.occlusion.bias", (1,))]
    return out


def _strip(entries, prefix):
    return [(n[len(prefix):], shp) for n, shp in entries if n.startswith(prefix)]


def kp_detector_manifest(common_params, kp_detector_params):
    """[(name, shape)] of a standalone KPDetector (archs/keypoint_detector_arch.py:13-47): the kp_detector.* entries."""
    dm_stub = {"block_expansion": 64, "max_features": 1024, "num_blocks": 5}
    return _strip(motion_estimator_manifest(common_params, dm_stub, kp_detector_params), "kp_detector.")


def dense_motion_manifest(common_params, dense_motion_params):
    """[(name, shape)] of a standalone DenseMotionNetwork (archs/dense_motion_arch.py:12-63)."""
    kp_stub = {"block_expansion": 32, "max_features": 1024, "num_blocks": 5}
    return _strip(motion_estimator_manifest(common_params, dense_motion_params, kp_stub), "dense_motion_network.")
