#!/bin/bash
PKG=synergize_motion_appearance_amd
cp $PKG/lib/gemm_rp_bf3.o /tmp/gemm_rp_bf3.o.keep
BASE="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-comment -mllvm -amdgpu-mfma-vgpr-form -Xclang -target-feature -Xclang -packed-fp32-ops -I include -I $PKG/csrc"
for V in "$@"; do
  hipcc $BASE $V -c $PKG/csrc/gemm_rp_bf3.hip -o $PKG/lib/gemm_rp_bf3.o 2>/dev/null || { echo "variant '$V': compile failed"; continue; }
  hipcc --offload-arch=gfx950 -shared -fPIC $PKG/lib/*.o -o $PKG/lib/libsmx.so
  echo "== variant '$V'"; timeout 200 python tools/gemm_bf3_bench.py 2>&1 | grep -v amdgpu.ids | awk 'NR>1{printf "%s/%s ", $8, $10} END{print ""}'
done
cp /tmp/gemm_rp_bf3.o.keep $PKG/lib/gemm_rp_bf3.o
hipcc --offload-arch=gfx950 -shared -fPIC $PKG/lib/*.o -o $PKG/lib/libsmx.so
