#!/usr/bin/env python3
"""Single-process reproducer for the shared-device anomaly: render ONE batch over and over and compare every
pass with the first one bit for bit, while something else (tools/cwsr_probe antagonist ..., or a second copy of this script) uses
the same GPU.  With --trace every ops.* call of every pass is fingerprinted (inputs before, outputs after: int64 sums of the raw
bit patterns, kept on the device and read back once per pass); on a mismatching pass the first differing record names the launch.

usage: python tools/preempt_repro.py [--dtype bf16|f32] [--batch 6] [--passes 300] [--trace] [--tag T] [--seconds S]"""
import argparse
import os
import sys
import time

import torch
import yaml

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from basicsr.archs import build_network  # noqa: E402
from synergize_motion_appearance_amd import driver, ops  # noqa: E402
from synergize_motion_appearance_amd.synth import synth_state_dict, synth_clip  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--batch", type=int, default=6)
ap.add_argument("--passes", type=int, default=300)
ap.add_argument("--seconds", type=float, default=0.0, help="stop after this many seconds (0 = run all passes)")
ap.add_argument("--trace", action="store_true")
ap.add_argument("--tag", default="repro")
ap.add_argument("--reencode", action="store_true", help="re-encode the source state before every pass (the bench's timed prologue)")
ap.add_argument("--poison", type=int, default=0, help="bit 1: fill every CU's LDS with NaN before every C-ABI call, bit 2: every SIMD's VGPRs/AGPRs "
                                                      "(tools/poison.hip -> /tmp/libpoison.so); the first two passes stay clean (the reference)")
ap.add_argument("--poison-pattern", default="0xFFFFFFFF")
ap.add_argument("--keep", type=int, default=0, help="keep full copies of the first K fingerprinted tensors of every pass; a mismatching pass is compared with the "
                                                    "reference pass element by element (a sum fingerprint cannot see a permutation)")
args = ap.parse_args()

cfg = yaml.safe_load(open(os.path.join(REPO, "options/test.yml")))
net_g, me = build_network(cfg["network_g"]), build_network(cfg["network_motion_estimator"])
net_g.load_state_dict(synth_state_dict([(k, v.shape) for k, v in net_g.state_dict().items()]), strict=True)
me.load_state_dict(synth_state_dict([(k, v.shape) for k, v in me.state_dict().items()]), strict=True)
net_g, me = net_g.cuda().eval(), me.cuda().eval()
net_g.set_compute_dtype(args.dtype)
me.set_compute_dtype(args.dtype)
src, drv = synth_clip(args.batch + 1, seed=31)
src, drv = src.cuda(), drv.cuda()

RECORDS = None          # list of (label, device int64 scalar) of the current pass
KEPT = []               # --keep: clones of the first K fingerprinted tensors of the current pass
POISON = [0]            # what the C-ABI proxy poisons before each call (0 = nothing)
if args.poison:
    import ctypes as C
    from synergize_motion_appearance_amd import lib as L
    _pl = C.CDLL("/tmp/libpoison.so")
    _pl.poison_cu_state.restype, _pl.poison_cu_state.argtypes = C.c_int, [C.c_uint, C.c_int, C.c_void_p, C.c_void_p]
    _sink = torch.zeros(4, dtype=torch.int32, device="cuda")
    _real = L.load()
    _pat = int(args.poison_pattern, 0)

    class _Proxy:
        def __getattr__(self, name):
            fn = getattr(_real, name)
            if not name.startswith("smx_"):
                return fn

            def call(*a):
                if POISON[0]:
                    rc = _pl.poison_cu_state(_pat, POISON[0], _sink.data_ptr(), torch.cuda.current_stream().cuda_stream)
                    assert rc == 0, rc
                return fn(*a)
            return call
    L._lib = _Proxy()


def _bits(t):
    if t.dtype == torch.float32:
        return t.view(torch.int32)
    if t.dtype in (torch.bfloat16, torch.float16):
        return t.view(torch.int16)
    if t.dtype in (torch.uint8, torch.int8, torch.int32, torch.int64, torch.int16):
        return t
    return None


def _tensors(x, prefix, out):
    if torch.is_tensor(x):
        if x.is_cuda and x.numel() > 0:
            out.append((prefix, x))
    elif isinstance(x, (list, tuple)):
        for i, y in enumerate(x):
            _tensors(y, f"{prefix}[{i}]", out)
    elif isinstance(x, dict):
        for k, y in x.items():
            _tensors(y, f"{prefix}.{k}", out)


def _record(name, phase, obj):
    ts = []
    _tensors(obj, "", ts)
    for label, t in ts:
        b = _bits(t)
        if b is None:
            continue
        RECORDS.append((f"{name}:{phase}{label} {tuple(t.shape)} {str(t.dtype)[6:]}", torch.sum(b, dtype=torch.int64)))
        if len(KEPT) < args.keep:
            KEPT.append(t.clone())


def _wrap(name, fn):
    def w(*a, **k):
        if RECORDS is None:
            return fn(*a, **k)
        _record(name, "in", (a, k))
        r = fn(*a, **k)
        _record(name, "out", r)
        _record(name, "arg-after", (a, k))          # in-place results (out= buffers, slices of a concat)
        return r
    return w


if args.trace:
    for name, fn in list(vars(ops).items()):
        if callable(fn) and getattr(fn, "__module__", None) == ops.__name__ and not name.startswith("_") and not isinstance(fn, type) \
                and name not in ("set_tuning", "profile"):
            setattr(ops, name, _wrap(name, fn))


def one_pass(state):
    global RECORDS
    if args.reencode:
        state = driver.encode_source_state(net_g, me, src, drv[0:1], True)
    RECORDS = [] if args.trace else None
    del KEPT[:]
    out = driver.render_frames(state, drv[1:1 + args.batch], net_g, me, True, True, batch=args.batch)
    rec, RECORDS = RECORDS, None
    if rec is not None:
        labels = [l for l, _ in rec]
        vals = torch.stack([v for _, v in rec]).cpu()
        return out, (labels, vals, list(KEPT))
    return out, None


def compare_kept(rec, ref_rec, limit=10):
    """element-wise comparison of the kept tensors of a pass with the reference pass: which records differ, where, and by what."""
    lines = []
    for i, (a, b) in enumerate(zip(rec[2], ref_rec[2])):
        if a.shape != b.shape or a.dtype != b.dtype:
            lines.append(f"    #{i} {rec[0][i]}: shape / dtype differs")
            continue
        ne = (_bits(a) != _bits(b))
        n = int(ne.sum())
        if n:
            idx = ne.nonzero()[:4].tolist()
            lead = sorted(set(ix[0] for ix in ne.nonzero()[:, :1].tolist())) if ne.dim() > 1 else []
            vals = [(tuple(ix), float(a[tuple(ix)].float()), float(b[tuple(ix)].float())) for ix in idx]
            lines.append(f"    #{i} {rec[0][i]}: {n} of {a.numel()} elements differ; leading indices {lead[:8]}; first (index, got, reference): {vals}")
            if len(lines) >= limit:
                break
    return "\n".join(lines) if lines else "    (all kept tensors equal the reference pass element by element)"


def explain(rec, ref_rec, stable):
    labels, vals = rec[0], rec[1]
    if labels != ref_rec[0]:
        return " | launch sequence differs from the first pass"
    diff = [i for i in (vals != ref_rec[1]).nonzero().flatten().tolist() if stable[i]]
    return f" | {len(diff)} of {len(labels)} fingerprints differ; first: " + " ;; ".join(f"#{i} {labels[i]}" for i in diff[:8])


with torch.no_grad():
    state = driver.encode_source_state(net_g, me, src, drv[0:1], True)
    ref, ref_rec = one_pass(state)
    ref2, ref2_rec = one_pass(state)
    torch.cuda.synchronize()
    # fingerprints that differ between two CLEAN passes are noise (an `out=` buffer hashed before it is written): masked below
    stable = (ref_rec[1] == ref2_rec[1]).tolist() if ref_rec else None
    print(f"[{args.tag}] {args.dtype} B={args.batch} trace={args.trace} poison={args.poison}: warm passes equal: {bool(torch.equal(ref, ref2))}; "
          f"{len(ref_rec[0]) if ref_rec else 0} fingerprints per pass ({sum(stable) if stable else 0} stable)", flush=True)
    POISON[0] = args.poison
    bad, t0, done, quiet = 0, time.time(), 0, 0
    for p in range(args.passes):
        out, rec = one_pass(state)
        done += 1
        if not torch.equal(out, ref):
            bad += 1
            d = (out.int() - ref.int()).abs()
            per_frame = [int(d[i].max()) for i in range(d.shape[0])]
            if bad <= 12:
                print(f"[{args.tag}] MISMATCH pass {p} t={time.time() - t0:.1f}s: per-frame max LSB {per_frame}, bytes differing {int((d > 0).sum())}"
                      + (explain(rec, ref_rec, stable) if rec is not None else ""), flush=True)
                if rec is not None and args.keep:
                    print(compare_kept(rec, ref_rec), flush=True)
        elif rec is not None and quiet < 3:
            diff = [i for i in (rec[1] != ref_rec[1]).nonzero().flatten().tolist() if stable[i]]
            if diff:
                quiet += 1
                print(f"[{args.tag}] pass {p}: output equal but {len(diff)} stable fingerprints differ; first: " + " ;; ".join(f"#{i} {rec[0][i]}" for i in diff[:4]), flush=True)
        if args.seconds and time.time() - t0 > args.seconds:
            break
    torch.cuda.synchronize()
    print(f"[{args.tag}] done: {done} passes in {time.time() - t0:.1f}s, {bad} mismatching", flush=True)
