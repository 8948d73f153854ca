#!/bin/bash
# s_memtime stamps of the split-bf16 Winograd kernel's block phases (a -DB3_ABL=64 build of csrc/winograd_bf3.hip only: the stamps overwrite
# the GroupNorm partials).  usage: tools/wino_bf3_trace.sh [extra ablation mask, or-ed with 64] [B]  -> stdout
set -u
MASK=$(( ${1:-0} | 64 ))
B=${2:-300}
PKG=synergize_motion_appearance_amd
cp $PKG/lib/winograd_bf3.o /tmp/winograd_bf3.o.keep
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-comment -mllvm -amdgpu-mfma-vgpr-form -Xclang -target-feature -Xclang -packed-fp32-ops -I include -I $PKG/csrc"
hipcc $FLAGS -DB3_ABL=$MASK -c $PKG/csrc/winograd_bf3.hip -o $PKG/lib/winograd_bf3.o 2>/dev/null
hipcc --offload-arch=gfx950 -shared -fPIC $PKG/lib/*.o -o $PKG/lib/libsmx.so
python - "$B" "$MASK" <<'PY'
import sys, torch
import numpy as np
sys.path.insert(0, ".")
from synergize_motion_appearance_amd import ops
B, MASK = int(sys.argv[1]), int(sys.argv[2])
ops.WINO_BF3 = 6; ops.WINO_BF3_MIN_BLOCKS = 1
print(f"mask {MASK}: per block (wave 0..7 mean), shader cycles: prologue (entry -> loop) | main loop | epilogue | total ; launch us")
for cin, cout, s in ((128, 128, 128), (64, 64, 256), (256, 128, 64), (512, 256, 32)):
    x = torch.randn((B, s, s, cin), device="cuda")
    cv = ops.Conv.from_torch(torch.randn((cout, cin, 3, 3), device="cuda") / (3 * cin ** 0.5), torch.randn(cout, device="cuda") * 0.1)
    out = torch.empty((B, s, s, cout), device="cuda"); res = torch.randn((B, s, s, cout), device="cuda")
    ss = torch.stack([1 + 0.2 * torch.rand((B, cin), device="cuda"), 0.1 * torch.randn((B, cin), device="cuda")], -1).contiguous()
    for _ in range(2):
        y = ops.conv(x, cv, out=out, in_ss=ss, in_swish=True, res=res, want_stats=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); y = ops.conv(x, cv, out=out, in_ss=ss, in_swish=True, res=res, want_stats=True); e1.record(); torch.cuda.synchronize()
    nblk = B * (s // 16) * (s // 16) * (cout // 64)
    part = y._gn_part.view(-1).view(torch.int64)[: nblk * 64].view(nblk, 8, 8).cpu().numpy()
    t = part[:, :, :4].astype(np.float64)
    d = np.diff(t, axis=2)                      # [blocks][waves][4]
    tot = t[:, :, 3] - t[:, :, 0]
    span = (t[:, :, 3].max() - t[:, :, 0].min())
    print(f"{cin}->{cout}@{s}: " + " | ".join(f"{d[:, :, k].mean():9.0f}" for k in range(3)) + f" | {tot.mean():9.0f} ; {1e3 * e0.elapsed_time(e1):8.1f} us; blocks/CU {nblk / 256:.0f}; effective clock {tot[:, 0].sum() / 256 / (1e3 * e0.elapsed_time(e1)) / 1e3:.2f} GHz (block totals / 256 CUs / launch time)")
    del x, out, res
PY
cp /tmp/winograd_bf3.o.keep $PKG/lib/winograd_bf3.o
hipcc --offload-arch=gfx950 -shared -fPIC $PKG/lib/*.o -o $PKG/lib/libsmx.so
