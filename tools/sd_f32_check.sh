#!/bin/bash
# fp32 pipeline (packed-fp32 Winograd kernels) bit-stable next to another process's bf16 MFMA convolution?
export HSA_ENABLE_IPC_MODE_LEGACY=0
python tools/aggressor.py t32 60 > /tmp/agg.log 2>&1 &
APID=$!
sleep 25
timeout 120 python tools/preempt_repro.py --dtype f32 --passes 1000000 --seconds 25 --tag f32-next-to-t32 2>&1 | grep -v amdgpu.ids | tail -4
kill $APID 2>/dev/null; wait $APID 2>/dev/null
tail -2 /tmp/agg.log
