#!/usr/bin/env python3
"""Time the implicit-GEMM conv kernel per (shape, tile config) on the GPU box.
usage: python tools/gemm_tune.py [out.txt]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from synergize_motion_appearance_amd import ops  # noqa: E402

# (B, H, Cin, Cout, k) conv shapes that dominate the step (B = 10 frames)
SHAPES = [(10, 128, 128, 128, 3), (10, 256, 64, 64, 3), (10, 64, 256, 256, 3), (10, 32, 256, 512, 3), (10, 32, 512, 256, 3),
          (10, 256, 128, 64, 3), (10, 256, 128, 128, 3), (10, 32, 256, 256, 3), (10, 32, 256, 256, 1), (10, 64, 128, 128, 3),
          (10, 64, 192, 128, 3), (10, 64, 256, 128, 3), (10, 256, 64, 192, 1), (10, 32, 256, 512, 1), (10, 64, 128, 96, 3)]
TILES = [0]


def main():
    out = open(sys.argv[1], "w") if len(sys.argv) > 1 else sys.stdout
    out.write("B H Cin Cout k | " + " ".join(f"tile{t:>2}" for t in TILES) + "   (TFLOP/s; 0 = rejected)\n")
    for (B, H, Cin, Cout, k) in SHAPES:
        x = torch.randn(B, H, H, Cin, device="cuda")
        cv = ops.Conv.from_torch(torch.randn(Cout, Cin, k, k, device="cuda") * 0.05, torch.randn(Cout, device="cuda"))
        y = torch.empty(B, H, H, Cout, device="cuda")
        fl = 2.0 * B * H * H * Cout * Cin * k * k
        row = []
        for t in TILES:
            try:
                for _ in range(2):
                    ops.conv(x, cv, out=y, tile=t)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                n = 5
                for _ in range(n):
                    ops.conv(x, cv, out=y, tile=t)
                e1.record()
                torch.cuda.synchronize()
                row.append(fl / (e0.elapsed_time(e1) / n * 1e-3) / 1e12)
            except Exception:
                row.append(0.0)
        out.write(f"{B} {H} {Cin} {Cout} {k} | " + " ".join(f"{v:6.1f}" for v in row) + "\n")
        out.flush()


if __name__ == "__main__":
    main()
