#!/usr/bin/env python3
"""Timing-only ablations of the split Winograd kernel (SMX_TOOLS build): executed MFMA fraction with one ingredient of the MFMA waves' loop removed.
usage: SMX_TOOLS=1 python -m synergize_motion_appearance_amd.build && SMX_TOOLS=1 python tools/wino_ws_ablate.py [B]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from synergize_motion_appearance_amd import ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 300
ABL = [(0, "full"), (1, "-transform"), (2, "-U"), (3, "-transform -U"), (4, "-slot handshake"), (7, "-tr -U -hs"), (8, "-handover"), (15, "all off"), (20, "no helpers"), (21, "no helpers -tr"), (22, "no helpers -U"), (23, "no helpers -tr -U"), (31, "bare MFMA loop")]


def timed(fn, n=4):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


print("executed MFMA fraction of 157.3 TF: wide kernel | split kernel: " + " | ".join(n for _, n in ABL))
for cin, cout, s in [(64, 64, 256), (128, 128, 128), (256, 128, 64), (512, 256, 32)]:
    x = torch.randn((B, s, s, cin), device="cuda")
    cv = ops.Conv.from_torch(torch.randn((cout, cin, 3, 3), device="cuda") / (3 * cin ** 0.5), torch.randn(cout, device="cuda") * 0.1)
    out, res, ss = torch.empty((B, s, s, cout), device="cuda"), torch.randn((B, s, s, cout), device="cuda"), torch.rand((B, cin, 2), device="cuda")
    fl = 2.0 * B * s * s * cout * 9 * cin
    run = lambda: ops.conv(x, cv, out=out, in_ss=ss, in_swish=True, res=res, want_stats=True)   # noqa: E731
    ops.set_tuning("wino_ws", 0)
    row = [f"{fl * 4 / 9 / timed(run) / 1e9 / 157.3:.3f}"]
    ops.set_tuning("wino_ws", 1)
    for a, _ in ABL:
        ops.set_tuning("wino_ablate", a)
        row.append(f"{fl * 4 / 9 / timed(run) / 1e9 / 157.3:.3f}")
    ops.set_tuning("wino_ablate", 0)
    print(f"{cin:4d}->{cout:4d} @{s:3d}: " + " | ".join(row), flush=True)
    plain = lambda: ops.conv(x, cv, out=out)   # noqa: E731  no GroupNorm loader, no residual, no partials: light helper waves
    row = []
    for a in (0, 7):
        ops.set_tuning("wino_ablate", a)
        row.append(f"{fl * 4 / 9 / timed(plain) / 1e9 / 157.3:.3f}")
    ops.set_tuning("wino_ablate", 0)
    ops.set_tuning("wino_ws", 0)
    row.append(f"{fl * 4 / 9 / timed(plain) / 1e9 / 157.3:.3f}")
    print(f"        plain conv (light helpers): split full | split -tr -U -hs | wide: " + " | ".join(row), flush=True)
