#!/bin/bash
# rebuild only attention.o with each flag set, run the debug comparison, restore
PKG=synergize_motion_appearance_amd
cp $PKG/lib/attention.o /tmp/attention.o.keep
BASE="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-comment -Xclang -target-feature -Xclang -packed-fp32-ops -I include -I $PKG/csrc"
for V in "$@"; do
  hipcc $BASE $V -c $PKG/csrc/attention.hip -o $PKG/lib/attention.o 2>/dev/null || { echo "variant '$V': compile failed"; continue; }
  hipcc --offload-arch=gfx950 -shared -fPIC $PKG/lib/*.o -o $PKG/lib/libsmx.so
  echo "== variant '$V'"; timeout 200 python tools/scratch/attn_dbg.py 2>&1 | grep -v amdgpu.ids
done
cp /tmp/attention.o.keep $PKG/lib/attention.o
hipcc --offload-arch=gfx950 -shared -fPIC $PKG/lib/*.o -o $PKG/lib/libsmx.so
