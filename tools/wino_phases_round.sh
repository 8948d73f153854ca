#!/bin/bash
# The fp32 Winograd kernel's phase evidence at the bench batch (B = 300 frames per step): timing-only ablations + per-wave s_memtime phase table.
# Needs the -DSMX_TOOLS build of the library (rebuilt here, on the GPU box; the shipped library rejects wino_ablate != 0).
OUT=$PWD/gpurun_out; mkdir -p $OUT
SMX_TOOLS=1 python -c "from synergize_motion_appearance_amd import build; build.build(force=True, verbose=False)" > /dev/null 2>&1; echo "tools build rc=$?"
{
echo "# Round 4: where the wide Winograd kernel's time goes at B = 300 (the bench's frames per step).  MI355X, ResBlock-form launches (GN+swish loader, residual, stats)."
echo "# Commands: SMX_TOOLS=1 build; python tools/wino_bench.py 300 ablate ; python tools/wino_trace.py 300 128 128 256 ; python tools/wino_trace.py 300 64 64 256 ; python tools/wino_trace.py 300 256 256 32"
echo; echo "## 1. Ablations (timing-only builds: one phase compiled out; outputs are wrong by construction)"
SMX_TOOLS=1 python tools/wino_bench.py 300 ablate 2>&1 | grep -v amdgpu.ids
for a in "128 128 256" "64 64 256" "256 256 32"; do echo; echo "## per-wave phase timeline (s_memtime stamps, wino_ablate=32), $a"; SMX_TOOLS=1 python tools/wino_trace.py 300 $a 2>&1 | grep -v amdgpu.ids; done
} > $OUT/wino_phases_r04.txt
tail -5 $OUT/wino_phases_r04.txt
