#!/bin/bash
# A/B of compile-time variants of csrc/winograd_bf3.hip on the GPU box: rebuilds ONLY that object with each -D set, times three shapes (ResBlock form), restores the library.
# usage: tools/wino_bf3_variant.sh "<-D flags of variant 1>" "<variant 2>" ...      (an empty string = the shipped build)
PKG=synergize_motion_appearance_amd
cp $PKG/lib/winograd_bf3.o /tmp/winograd_bf3.o.keep
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-comment -mllvm -amdgpu-mfma-vgpr-form -Xclang -target-feature -Xclang -packed-fp32-ops -I include -I $PKG/csrc"
for V in "$@"; do
  hipcc $FLAGS $V -c $PKG/csrc/winograd_bf3.hip -o $PKG/lib/winograd_bf3.o 2>/dev/null || { echo "variant '$V': compile failed"; continue; }
  hipcc --offload-arch=gfx950 -shared -fPIC $PKG/lib/*.o -o $PKG/lib/libsmx.so
  python - "$V" <<'PY'
import sys, torch
sys.path.insert(0, ".")
from synergize_motion_appearance_amd import ops
ops.WINO_BF3 = 6; ops.WINO_BF3_MIN_BLOCKS = 1
B = 300
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
    row.append(f"{cin}->{cout}@{s}: {1e3 * e0.elapsed_time(e1) / 5:8.1f}")
    del x, out, res
print(f"variant '{sys.argv[1]}' | " + " | ".join(row), flush=True)
PY
done
cp /tmp/winograd_bf3.o.keep $PKG/lib/winograd_bf3.o
hipcc --offload-arch=gfx950 -shared -fPIC $PKG/lib/*.o -o $PKG/lib/libsmx.so
