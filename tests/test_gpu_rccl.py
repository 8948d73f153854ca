"""RCCL on the one GPU of the test box (round-4 review: "RCCL itself has never carried a byte").  A process group of world_size 1 over
the `nccl` backend (= RCCL on ROCm) runs every collective call site of the multi-GPU path -- the communicator is real, the kernels
are RCCL's, the stream semantics (`async_op` work handles joined with `wait()`, collectives ordered behind the launch stream) are
the ones an 8-GPU job uses; only the peer is missing:

  * `driver.broadcast_source_states` (async broadcasts of the packed source state, waited together) and `animate_sharded(gather=True)`
    through the DEVICE gather branch (`driver.py`: the host staging is gloo-only), against `animate_batched`;
  * `TrainStep` with its collectives forced on at world 1: construction-time broadcast of parameters / Adam state / BatchNorm buffers,
    the flat-gradient all-reduce issued from inside the backward (`overlap_allreduce`), eager and through the two-hipGraph form, against
    a TrainStep that never touches torch.distributed (bit-identical parameters after the steps);
  * `bench.py --gpus 1` under `torch.distributed.run` (no SMX_BENCH_BACKEND): RCCL initialised, the source state broadcast inside the
    timed region, `config.parallelism` says so.
Reference: `/root/reference/basicsr/utils/dist_util.py:10-57` (init), `/root/reference/basicsr/models/base_model.py:71-74` (DDP)."""
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


def _render_worker(port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
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
        src, drv = synth_clip(9, seed=31)
        src, drv = src.cuda(), drv.cuda()
        res = {"backend": dist.get_backend()}
        for dt in ("f32", "bf16"):
            net_g.set_compute_dtype(dt)
            me.set_compute_dtype(dt)
            with torch.no_grad():
                one = driver.animate_batched(src, drv, net_g, me, True, True, batch=4, anchor_idx=2)
                out = driver.animate_sharded(src, drv, net_g, me, relative=True, adapt_movement_scale=True, batch=4, root=0, anchor_idx=2, gather=True)
                # several sources: every owner encodes first, then all broadcasts in flight at once, waited together
                src2 = synth_clip(1, seed=77)[0].cuda()
                states = driver.broadcast_source_states(net_g, me, {0: (src.unsqueeze(0), drv[2:3]), 1: (src2.unsqueeze(0), drv[2:3])}, {0: 0, 1: 0}, True)
                again = driver.render_frames(states[0], drv, net_g, me, True, True, batch=4)
                other = driver.render_frames(states[1], drv, net_g, me, True, True, batch=4)
            torch.cuda.synchronize()
            res[dt] = {"gather_vs_single": int((out.int() - one.int()).abs().max()), "bcast_vs_single": int((again.int() - one.int()).abs().max()),
                       "other_source_differs": int((other.int() - one.int()).abs().max()), "shape": tuple(out.shape), "is_cuda": bool(out.is_cuda)}
        q.put(res)
    finally:
        dist.barrier()
        dist.destroy_process_group()


def _run(target, timeout=900):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=target, args=(_free_port(), q))
    p.start()
    import queue
    import time
    t0, res = time.time(), None
    while res is None:
        try:
            res = q.get(timeout=5)
        except queue.Empty:
            assert p.is_alive() or not q.empty(), f"the worker died (exit code {p.exitcode}) before it reported"
            assert time.time() - t0 < timeout, "the worker did not report in time"
    p.join(timeout=120)
    assert p.exitcode == 0
    return res


def test_rccl_world1_broadcast_and_device_gather_equal_the_single_process_frames():
    assert torch.cuda.is_available(), "needs an MI355X"
    res = _run(_render_worker)
    assert res["backend"] == "nccl"
    for dt in ("f32", "bf16"):
        r = res[dt]
        assert r["shape"] == (9, 256, 256, 3) and r["is_cuda"]
        # same batching on both sides: the broadcast state is a bit copy of the encoded one, so the frames are identical
        assert r["gather_vs_single"] == 0 and r["bcast_vs_single"] == 0, (dt, r)
        assert r["other_source_differs"] > 8, (dt, r)


def _train_worker(port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)
    import yaml
    from basicsr.archs import build_network
    from synergize_motion_appearance_amd.synth import synth_state_dict, synth_clip
    from synergize_motion_appearance_amd.trainer import TrainStep, EquivarianceTransform
    cfg = yaml.safe_load(open(os.path.join(REPO, "options/train.yml")))
    topt = {k: v for k, v in cfg["train"].items() if k not in ("perceptual_opt", "gan_opt", "kp_distance_opt")}
    _, clip = synth_clip(8, seed=321)
    src, drv = clip[[0, 5]].contiguous().cuda(), clip[[3, 7]].contiguous().cuda()

    def fresh(use_graph, collectives):
        net_g, me = build_network(cfg["network_g"]), build_network(cfg["network_motion_estimator"])
        net_g.load_state_dict(synth_state_dict([(k, v.shape) for k, v in net_g.state_dict().items()]), strict=True)
        me.load_state_dict(synth_state_dict([(k, v.shape) for k, v in me.state_dict().items()]), strict=True)
        TrainStep.COLLECTIVES_AT_WORLD_1 = collectives
        return TrainStep(net_g.cuda(), me.cuda(), topt, use_graph=use_graph)

    def run(use_graph, collectives):
        step = fresh(use_graph, collectives)
        init = step.g.flat.value.clone()
        gen = torch.Generator().manual_seed(11)
        for _ in range(4):                                     # graph form: 2 eager steps, capture + replay, replay
            tf = EquivarianceTransform(2, sigma_affine=0.05, sigma_tps=0.005, points_tps=5, generator=gen)
            losses, _ = step.step(src, drv, transform=tf)
        torch.cuda.synchronize()
        return step.g.flat.value.clone(), init, float(losses["l_g_total"]), (step._graph2 is not None)
    base = run(False, False)                                   # never touches torch.distributed
    base2 = run(False, False)                                  # ... twice: the run-to-run noise of the warp-backward atomics under Adam's g / sqrt(v)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        # (i) the collectives themselves are identities at world 1, bit for bit, through RCCL
        step = fresh(False, True)
        before = [f.value.clone() for f in (step.g.flat, step.flat_m)]
        synced = step.sync_replicas()
        same_params = all(bool(torch.equal(b, f.value)) for b, f in zip(before, (step.g.flat, step.flat_m)))
        tf = EquivarianceTransform(2, sigma_affine=0.05, sigma_tps=0.005, points_tps=5, generator=torch.Generator().manual_seed(11))
        step.g.flat.zero_grad(), step.flat_m.zero_grad()
        step.forward_backward(src, drv, transform=tf)
        g0, m0 = step.g.flat.grad.clone(), step.flat_m.grad.clone()
        pend = step.g.flat.all_reduce_start(dist) + step.flat_m.all_reduce_start(dist)
        n_async = len(pend)
        step.g.flat.all_reduce_wait(pend)
        torch.cuda.synchronize()
        same_grads = bool(torch.equal(g0, step.g.flat.grad)) and bool(torch.equal(m0, step.flat_m.grad))
        # (ii) whole steps with the collectives in them: eager, and replayed from the two hipGraphs with the all-reduce between them
        eager = run(False, True)
        graph = run(True, True)
        TrainStep.COLLECTIVES_AT_WORLD_1 = False
        move = float((base[0] - base[1]).abs().mean())
        q.put({"synced": bool(synced), "same_params": same_params, "same_grads": same_grads, "async_works": n_async,
               "noise": float((base2[0] - base[0]).abs().mean()) / move, "eager": float((eager[0] - base[0]).abs().mean()) / move,
               "graph": float((graph[0] - base[0]).abs().mean()) / move, "two_graphs": graph[3], "loss": (base[2], base2[2], eager[2], graph[2]),
               "finite": bool(torch.isfinite(eager[0]).all() and torch.isfinite(graph[0]).all())})
    finally:
        dist.barrier()
        dist.destroy_process_group()


def test_rccl_world1_training_step_collectives():
    """the construction-time broadcast and the flat-gradient all-reduce over RCCL: at world 1 both are identities -- checked bit for bit on the
    buffers themselves -- and whole steps with the collectives inside (eager; and replayed from TWO hipGraphs with net_g's all-reduce issued
    between them) land where steps without any collective land, to the run-to-run noise two collective-free runs show among themselves (the
    warp-backward atomics, amplified by Adam's g / sqrt(v) in the first steps)."""
    assert torch.cuda.is_available(), "needs an MI355X"
    r = _run(_train_worker, timeout=1500)
    assert r["synced"] and r["same_params"] and r["same_grads"] and r["async_works"] >= 2 and r["finite"], r
    assert r["two_graphs"] is True                            # the overlapped form: [.. backward of net_g] | all-reduce | [backward of the estimator]
    assert r["eager"] <= 2.0 * r["noise"] + 0.02 and r["graph"] <= 2.0 * r["noise"] + 0.02, r
    spread = abs(r["loss"][1] - r["loss"][0])                   # two collective-free runs: what the noise does to the fourth step's loss
    assert all(abs(l - r["loss"][0]) <= 3.0 * spread + 2e-2 * abs(r["loss"][0]) for l in r["loss"]), r


def test_bench_one_rank_under_torchrun_uses_rccl():
    """`bench.py --gpus 1` launched like the N>1 runs (torch.distributed.run, no backend override): the process group is RCCL, the packed
    source state is broadcast inside the timed region, and the line says so."""
    assert torch.cuda.is_available(), "needs an MI355X"
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("SMX_BENCH_BACKEND", "SMX_BENCH_ONE_DEVICE"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(REPO, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
           "--batch", "12", "--no-cpu-baseline", "--no-roofline", "--no-bf16-leg", "--no-train-leg", "--no-d2h"]
    r = subprocess.run(cmd, cwd=REPO, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert j["n_gpus"] == 1 and j["value"] > 0 and j["batch_consistency"]["max_lsb_vs_b1"] <= 1
    assert "RCCL" in j["config"]["parallelism"] and "gloo" not in j["config"]["parallelism"]
