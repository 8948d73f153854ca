#!/bin/bash
# Timing-only ablations of the split-bf16 Winograd kernel (csrc/winograd_bf3.hip, -DB3_ABL=mask: WRONG results by design), one rebuilt library per mask,
# on the GPU box.  usage: tools/wino_bf3_ablate.sh "0 1 2 4 8 16 ..." [B]   -> gpurun_out/bf3_ablate.txt
set -u
MASKS=${1:-"0 1 2 4 8 16"}
B=${2:-300}
export SMX_TOOLS=1
OUT=gpurun_out/bf3_ablate.txt
mkdir -p gpurun_out
echo "B3_ABL: 1 no transform/split, 2 no MFMAs, 4 no region staging, 8 no U requests, 16 no epilogue, 32 no scheduling fences   (us per launch, B=$B, ResBlock form)" > $OUT
for m in $MASKS; do
  SMX_HIPCC_EXTRA="-DB3_ABL=$m" python -c "from synergize_motion_appearance_amd import build; build.build(verbose=False)" >/dev/null 2>&1
  python - "$m" "$B" >> $OUT 2>&1 <<'PY'
import sys, torch
sys.path.insert(0, ".")
from synergize_motion_appearance_amd import ops
m, B = int(sys.argv[1]), int(sys.argv[2])
ops.WINO_BF3 = 6; ops.WINO_BF3_MIN_BLOCKS = 1
row = []
for cin, cout, s in ((128, 128, 128), (64, 64, 256), (256, 128, 64), (512, 256, 32)):
    x = torch.randn((B, s, s, cin), device="cuda")
    cv = ops.Conv.from_torch(torch.randn((cout, cin, 3, 3), device="cuda") / (3 * cin ** 0.5), torch.randn(cout, device="cuda") * 0.1)
    out = torch.empty((B, s, s, cout), device="cuda"); res = torch.randn((B, s, s, cout), device="cuda")
    ss = torch.stack([1 + 0.2 * torch.rand((B, cin), device="cuda"), 0.1 * torch.randn((B, cin), device="cuda")], -1).contiguous()
    f = lambda: ops.conv(x, cv, out=out, in_ss=ss, in_swish=True, res=res, want_stats=True)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): f()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 5
    fl = 2.0 * B * s * s * cout * 4 * cin * 6
    row.append(f"{cin}->{cout}@{s}: {1e3*t:8.1f} ({fl/t/1e9/2500:.3f})")
    del x, out, res
print(f"mask {m:3d} | " + " | ".join(row))
PY
done
SMX_HIPCC_EXTRA="" python -c "from synergize_motion_appearance_amd import build; build.build(verbose=False)" >/dev/null 2>&1
cat $OUT
