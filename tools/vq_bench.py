#!/usr/bin/env python3
"""A12 micro-benchmark: the fused nearest-codebook kernel (csrc/vq.hip) at the training-step shapes and the bench.py shape.
usage: python tools/vq_bench.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from synergize_motion_appearance_amd import ops  # noqa: E402

QUICK = bool(os.environ.get("VQ_QUICK"))          # only the bench.py shape (for variant sweeps)
for D in ((256,) if QUICK else (256, 32)):
    for N in ((1228800, 245760) if QUICK else (245760, 61440, 4096)):
        for Ks in ((1024,) if QUICK else (1024, 256)):
            z = torch.randn(N, D, device="cuda")
            cb = torch.randn(1024, D, device="cuda") / 32
            t0 = time.time()
            while time.time() - t0 < 0.2:                  # the clocks ramp over tens of ms: a cold first shape reads 10 % low
                ops.vq_nearest(z, cb, Ks)
                torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                ops.vq_nearest(z, cb, Ks)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 100
            print(f"D={D:3d} N={N:6d} Ks={Ks:4d}: {us:8.1f} us  {2.0 * N * Ks * D / us / 1e6:6.1f} TF (of 157.3)  "
                  f"{(4.0 * N * D * 2 + 8 * N) / us / 1e3:7.1f} GB/s algorithmic")
