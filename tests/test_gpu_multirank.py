"""The N>1 data path on hardware, before the 8-GPU scaling run sees it: two ranks (one process each, like
`torch.distributed.run`) share the ONE MI355X of the test box and run the real engines through
`driver.animate_sharded` -- rank 0 encodes the source, the packed state (encoder taps + down(source) + keypoints +
hull scale, 28.4 MB) is broadcast once, each rank renders its `shard_frames` block, rank 0 gathers -- and the result must
equal the 1-rank `animate_batched` frames (<= 1 LSB on uint8: the two runs batch the frames differently).
RCCL refuses two ranks on one device ("duplicate GPU"), so the process group here is gloo carrying device tensors;
`bench.py --gpus 2` is exercised the same way (SMX_BENCH_ONE_DEVICE=1, SMX_BENCH_BACKEND=gloo, stated in its JSON)."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_frames, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import yaml
        from basicsr.archs import build_network
        from synergize_motion_appearance_amd import driver
        from synergize_motion_appearance_amd.synth import synth_state_dict, synth_clip
        cfg = yaml.safe_load(open(os.path.join(REPO, "options/test.yml")))
        net_g, me = build_network(cfg["network_g"]), build_network(cfg["network_motion_estimator"])
        net_g.load_state_dict(synth_state_dict([(k, v.shape) for k, v in net_g.state_dict().items()]), strict=True)
        me.load_state_dict(synth_state_dict([(k, v.shape) for k, v in me.state_dict().items()]), strict=True)
        net_g, me = net_g.cuda().eval(), me.cuda().eval()
        src, drv = synth_clip(n_frames, seed=31)
        drv = drv.cuda()
        # only the root is given the source image: the other rank must live off the broadcast
        source = src.cuda() if rank == 0 else torch.full_like(src, float("nan")).cuda()
        out = driver.animate_sharded(source, drv, net_g, me, relative=True, adapt_movement_scale=True, batch=4, root=0,
                                     anchor_idx=2, gather=True)
        span, mine = driver.animate_sharded(source, drv, net_g, me, batch=3, root=0, anchor_idx=2, gather=False)
        res = {"rank": rank, "span": span, "mine_shape": tuple(mine.shape)}
        if rank == 0:
            one = driver.animate_batched(src.cuda(), drv, net_g, me, True, True, batch=5, anchor_idx=2)
            d = (out.int() - one.int()).abs()
            res.update(shape=tuple(out.shape), max_lsb=int(d.max()), frac_diff=float((d > 0).float().mean()),
                       shard_equal=int((mine.int() - one[span[0]:span[1]].int()).abs().max()))
        else:
            assert out is None
        q.put(res)
    finally:
        dist.barrier()
        dist.destroy_process_group()


def test_two_ranks_one_device_sharded_animation_equals_single_rank():
    assert torch.cuda.is_available(), "needs an MI355X"
    world, n_frames = 2, 13                                # ragged shards: 7 + 6
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_frames, q)) for r in range(world)]
    [p.start() for p in procs]
    res = sorted((q.get(timeout=600) for _ in range(world)), key=lambda r: r["rank"])
    [p.join(timeout=120) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    r0, r1 = res
    assert r0["span"] == (0, 7) and r1["span"] == (7, 13) and r0["mine_shape"] == (7, 256, 256, 3) and r1["mine_shape"] == (6, 256, 256, 3)
    assert r0["shape"] == (n_frames, 256, 256, 3)
    assert r0["max_lsb"] <= 1 and r0["shard_equal"] <= 1, r0
    assert r0["frac_diff"] < 0.01, r0                     # rounding ties only


@pytest.mark.parametrize("extra,scaling", [((), "weak"), (("--dtype", "bf16"), "weak"), (("--strong",), "strong")])
def test_bench_two_ranks_control_flow_on_one_device(tmp_path, extra, scaling):
    """`bench.py --gpus 2` exactly as the driver launches it (torch.distributed.run, one rank per process), both ranks on
    the one device of this box: the N>1 branch (owners encode first, async state broadcasts, window sharding) executes and
    prints a well-formed line -- in fp32, in bf16 (BASELINE configs[2] is the bf16 multi-GPU config: the broadcast carries
    raw bf16 bits) and in the strong-scaling form (ONE source, each step's frames split over the ranks).
    The number is NOT a scaling measurement (two ranks share one GPU)."""
    assert torch.cuda.is_available(), "needs an MI355X"
    env = dict(os.environ, SMX_BENCH_ONE_DEVICE="1", SMX_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--batch", "6", "--no-cpu-baseline", "--no-roofline", *extra]
    r = subprocess.run(cmd, cwd=REPO, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    j = json.loads(line)
    assert j["n_gpus"] == 2 and j["steps"] == 2 and j["scaling"] == scaling and j["value"] > 0
    if scaling == "weak":
        assert j["config"]["frames_total"] == 2 * 2 * 6 and j["config"]["sources"] == 2
    else:
        assert j["config"]["frames_total"] == 2 * 6 and j["config"]["sources"] == 1
    assert j["dtype"] == ("bf16" if "bf16" in extra else "f32")
    assert "gloo" in j["config"]["parallelism"] and j["config"]["one_device_test_knob"] is True
    assert len(j["rank_times_s"]["per_rank"]) == 2 and j["rank_times_s"]["max"] >= j["rank_times_s"]["min"] > 0
    if "bf16" in extra:
        assert j["batch_consistency"]["mean_lsb_vs_b1"] < 1.5
    else:
        assert j["batch_consistency"]["max_lsb_vs_b1"] <= 1


def test_bench_refuses_a_silent_backend_fallback():
    """RCCL cannot put two ranks on one device; without SMX_BENCH_BACKEND=gloo the bench must fail loudly on every
    rank (non-zero exit, no JSON line) instead of quietly measuring a host-staged broadcast."""
    assert torch.cuda.is_available(), "needs an MI355X"
    env = dict(os.environ, SMX_BENCH_ONE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0", SMX_BENCH_INIT_TIMEOUT_S="60")
    env.pop("SMX_BENCH_BACKEND", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
           "--batch", "2", "--no-cpu-baseline", "--no-roofline"]
    r = subprocess.run(cmd, cwd=REPO, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]


def _ddp_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import yaml
        from basicsr.archs import build_network
        from synergize_motion_appearance_amd.synth import synth_state_dict, synth_clip
        from synergize_motion_appearance_amd.trainer import TrainStep, EquivarianceTransform
        cfg = yaml.safe_load(open(os.path.join(REPO, "options/train.yml")))
        net_g, me = build_network(cfg["network_g"]), build_network(cfg["network_motion_estimator"])
        net_g.load_state_dict(synth_state_dict([(k, v.shape) for k, v in net_g.state_dict().items()]), strict=True)
        me.load_state_dict(synth_state_dict([(k, v.shape) for k, v in me.state_dict().items()]), strict=True)
        if rank == 1:                                          # a replica that starts elsewhere (BasicSR seeds rank r with seed + r): the constructor's
            with torch.no_grad():                              # rank-0 broadcast (DDP's, models/base_model.py:71-74) must bring it back
                net_g.state_dict()["generator.blocks.18.weight"].mul_(1.5)
                me.state_dict()["kp_detector.kp.weight"].add_(0.25)
                me.state_dict()["kp_detector.predictor.encoder.down_blocks.0.norm.running_mean"].add_(3.0)
        net_g, me = net_g.cuda(), me.cuda()
        topt = {k: v for k, v in cfg["train"].items() if k not in ("perceptual_opt", "gan_opt", "kp_distance_opt")}
        step = TrainStep(net_g, me, topt)
        start = [float(step.g.flat.value.double().sum()), float(step.flat_m.value.double().sum()),
                 float(step.bufs["kp_detector.predictor.encoder.down_blocks.0.norm.running_mean"].double().sum())]
        _, clip = synth_clip(8, seed=321)
        # each rank owns ONE different (source, driving) pair (a DistributedSampler shard)
        src, drv = clip[[0, 5][rank]][None].cuda(), clip[[3, 7][rank]][None].cuda()
        tf = EquivarianceTransform(1, sigma_affine=0.05, sigma_tps=0.005, points_tps=5, generator=torch.Generator().manual_seed(100 + rank))
        # local gradients first (no collective), then the real step
        step.g.flat.zero_grad(), step.flat_m.zero_grad()
        step.forward_backward(src, drv, transform=tf)
        local = [float(step.g.flat.grad.double().sum()), float(step.flat_m.grad.double().sum())]
        local_probe = step.g.flat.G["generator.blocks.18.weight"].clone()
        # BatchNorm running statistics moved in that probe pass: reset them so both passes start alike is NOT needed for this check
        step.g.flat.zero_grad(), step.flat_m.zero_grad()
        step.forward_backward(src, drv, transform=tf)
        step.g.flat.all_reduce(dist), step.flat_m.all_reduce(dist)
        summed = [float(step.g.flat.grad.double().sum()), float(step.flat_m.grad.double().sum())]
        summed_probe = step.g.flat.G["generator.blocks.18.weight"].clone()
        step.g.flat.adam_step(8e-5, gscale=1.0 / world)
        step.flat_m.adam_step(8e-5, gscale=1.0 / world)
        chk = [float(step.g.flat.value.double().sum()), float(step.flat_m.value.double().sum())]
        w_probe, wm_probe = step.g.flat.P["generator.blocks.18.weight"].cpu().numpy(), step.flat_m.P["kp_detector.kp.weight"].cpu().numpy()
        # then two whole steps through TrainStep.step: net_g's all-reduce is issued from inside the backward (overlap_allreduce)
        for _ in range(2):
            step.step(src, drv, transform=tf)
        torch.cuda.synchronize()
        after = [float(step.g.flat.value.double().sum()), float(step.flat_m.value.double().sum()), step.g.flat.t]
        # numpy, not tensors: a tensor in a Queue travels as a shared-memory handle that dies with this process
        q.put({"rank": rank, "local": local, "summed": summed, "start": start, "after": after, "local_probe": local_probe.cpu().numpy(), "summed_probe": summed_probe.cpu().numpy(),
               "w": w_probe, "wm": wm_probe, "chk": chk})
    finally:
        dist.barrier()
        dist.destroy_process_group()


def test_two_ranks_training_step_sums_gradients_and_keeps_replicas_identical():
    """SURVEY row N2 (DDP of models/base_model.py:71-74 as bucketed all-reduces of the flat gradient buffer): two ranks on the one
    device of this box (gloo, host-staged), each with its own (source, driving) pair: after the all-reduce both hold the SUM of the two
    local gradients, Adam applies it with gscale = 1/world (the average), and the replicas stay bit-identical."""
    assert torch.cuda.is_available(), "needs an MI355X"
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ddp_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in procs]
    res = sorted((q.get(timeout=900) for _ in range(world)), key=lambda r: r["rank"])
    [p.join(timeout=120) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    r0, r1 = res
    import numpy as np
    assert not np.array_equal(r0["local_probe"], r1["local_probe"])                    # different data, different local gradients
    assert np.array_equal(r0["summed_probe"], r1["summed_probe"])                      # one reduced gradient on both ranks
    ref = r0["local_probe"] + r1["local_probe"]
    assert float(np.abs(r0["summed_probe"] - ref).max()) < 1e-5 * float(np.abs(ref).max()) + 1e-9
    for i in range(2):
        assert abs(r0["summed"][i] - (r0["local"][i] + r1["local"][i])) < 1e-4 * (abs(r0["summed"][i]) + 1e-3)
    assert np.array_equal(r0["w"], r1["w"]) and np.array_equal(r0["wm"], r1["wm"]) and r0["chk"] == r1["chk"]   # replicas identical after Adam
    assert r0["start"] == r1["start"]                  # rank 1's perturbed parameters / BatchNorm buffer were replaced by rank 0's at construction
    assert r0["after"] == r1["after"] and r0["after"][2] == 3 and r0["after"][:2] != r0["chk"]          # and stay identical through TrainStep.step


@pytest.mark.parametrize("extra", [(), ("--eager", "--dtype", "f32"), ("--no-overlap",)])
def test_bench_train_two_ranks_on_one_device(extra):
    """`bench.py --train --gpus 2` (BASELINE configs[4] as the driver would launch it on an N-GPU node): two ranks on this box's one device, gloo
    carrying the flat buffers.  Default form = bf16 compute, the step replayed from TWO hipGraphs with net_g's gradient all-reduce issued between
    them; rank 1 starts from perturbed weights, so `replicas_bit_identical` also proves the construction-time broadcast."""
    assert torch.cuda.is_available(), "needs an MI355X"
    env = dict(os.environ, SMX_BENCH_ONE_DEVICE="1", SMX_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(REPO, "bench.py"), "--train", "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--batch", "1", *extra]
    r = subprocess.run(cmd, cwd=REPO, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert j["n_gpus"] == 2 and j["unit"] == "pairs/s" and j["value"] > 0 and j["replicas_bit_identical"] is True
    assert j["config"]["global_batch"] == 2 and j["dtype"] == ("f32" if "f32" in extra else "bf16")
    assert len(j["rank_times_s"]["per_rank"]) == 2
    if not extra:
        assert "two hipGraphs" in j["config"]["launch"]


def test_bench_512_two_ranks_on_one_device():
    """`bench.py --img-size 512 --gpus 2` (BASELINE configs[3] "1 then 8 GPUs"): the 512 variant through the same shard / broadcast path (113.5 MB of
    packed source state per broadcast)."""
    assert torch.cuda.is_available(), "needs an MI355X"
    env = dict(os.environ, SMX_BENCH_ONE_DEVICE="1", SMX_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(REPO, "bench.py"), "--img-size", "512", "--gpus", "2", "--steps", "1", "--warmup", "1",
           "--batch", "2", "--no-roofline"]
    r = subprocess.run(cmd, cwd=REPO, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert j["n_gpus"] == 2 and j["value"] > 0 and "512x512" in j["metric"] and j["config"]["sources"] == 2
    assert j["batch_consistency"]["max_lsb_vs_b1"] <= 1
