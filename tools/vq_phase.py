#!/usr/bin/env python3
"""Where does a vq_kernel block spend its life?  Needs a library built with SMX_TOOLS=1 (python -m synergize_motion_appearance_amd.build
with SMX_TOOLS=1 in the environment); SMX_VQ_PROF=1 then selects the instantiation that stamps s_memtime (the shader clock) at its
phase boundaries and leaves the five phase lengths in place of the first dmin values of every block.
usage: SMX_TOOLS=1 python -m synergize_motion_appearance_amd.build && SMX_VQ_PROF=1 python tools/vq_phase.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("SMX_VQ_PROF", "1")
from synergize_motion_appearance_amd import ops  # noqa: E402

D, Ks = 256, 1024
for N in (1228800, 245760):
    z = torch.randn(N, D, device="cuda")
    cb = torch.randn(1024, D, device="cuda") / 32
    for _ in range(3):
        idx, zq, dmin, sq = ops.vq_nearest(z, cb, Ks)
    torch.cuda.synchronize()
    raw = dmin.view(-1, 128)[:, :13].double().cpu()
    t = raw[:, :5]
    key = raw[:, 6].long()
    start = raw[:, 7] + raw[:, 8] * (1 << 24)
    end = raw[:, 9] + raw[:, 10] * (1 << 24)
    wall = (raw[:, 12] - raw[:, 11]) % (1 << 24)                      # 100 MHz ticks over the block's life
    mhz = ((end - start) / wall * 100)
    print(f"shader clock over a block's life: mean {mhz.mean():.0f} MHz  min {mhz.min():.0f}  max {mhz.max():.0f}")
    # per-CU timelines: residency and the gap between a block's end and the next start on the same CU
    gaps, busy, span = [], 0.0, 0.0
    for k in key.unique().tolist():
        m = key == k
        st, en = start[m], end[m]
        order = st.argsort()
        st, en = st[order], en[order]
        span += float(en.max() - st.min())
        busy += float((en - st).sum())
        ens = en.sort().values
        # the i-th start after the first `resident` ones follows the (i - resident)-th end
        res = 2
        if len(st) > res:
            gaps.append(st[res:] - ens[:len(st) - res])
    g = torch.cat(gaps)
    k0 = key.unique()[3]
    m = key == k0
    st0 = start[m].min()
    rows = sorted(zip((start[m] - st0).tolist(), t[m].tolist()))[:10]
    print("  one CU's first blocks: start | MFMA loop from .. to | end   (shader clocks since the CU's first block)")
    for st_, ph in rows:
        a = st_ + ph[0] + ph[1]
        print(f"    {st_:9.0f} | {a:9.0f} .. {a + ph[2]:9.0f} | {st_ + sum(ph):9.0f}")
    print(f"CUs seen {len(key.unique())}; mean blocks resident per CU over its span {busy / span:.2f}; "
          f"end -> next start on the same CU: mean {g.mean():.0f} clocks, median {g.median():.0f}, max {g.max():.0f}")
    names = ["z load + norms", "first code tile", "MFMA loop", "argmin over lanes", "gather + store"]
    tot = t.sum(1)
    ideal = 2.0 * (Ks // 32) * (D // 2) * 64            # two waves share a SIMD: 2 x tiles x MFMAs x 64 clocks
    print(f"N={N}: {t.shape[0]} blocks, mean block life {tot.mean():.0f} shader clocks ({tot.mean() / 2400:.1f} us at 2.4 GHz); "
          f"MFMA loop of a block with a partner on its SIMDs, back to back: {ideal:.0f}")
    for k, n in enumerate(names):
        print(f"  {n:18s} mean {t[:, k].mean():9.0f}  min {t[:, k].min():9.0f}  max {t[:, k].max():9.0f}   {100 * t[:, k].mean() / tot.mean():5.1f} %")
