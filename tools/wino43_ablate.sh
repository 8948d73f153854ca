#!/bin/bash
# timing-only ablations of the F(4x4,3x3) kernel (tools build: -DSMX_TOOLS): which phase carries the step time
export SMX_TOOLS=1
python -c "from synergize_motion_appearance_amd.build import build; build(force=True, verbose=False)" || exit 1
for abl in ${ABLS:-0 32 33 34 36 40 47 63}; do
  echo "== SMX_W43_ABL=$abl (1 no transform, 2 no MFMA, 4 no staging, 8 no U loads, 16 no barrier)"
  SMX_W43_ABL=$abl python tools/wino43_bench.py 60 2>/dev/null | sed -n '3,5p;9,10p' | awk '{print $1,$2,$3,":",$9,$10}'
done
