#!/usr/bin/env python3
"""Timing-only ablations of the 16x32-tile bf16 3x3 kernel (csrc/conv3x3_bf16_t32.hip built with -DSMX_TOOLS into tools/conv_t32_tools.bin by
`python tools/conv_t32_ablate.py --build`, which cross-compiles without a GPU): one phase compiled out per variant.
usage: python tools/conv_t32_ablate.py [B]"""
import ctypes as C
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
POL = os.environ.get("T32_RPOL", "")                       # region-request cache policy variant: "", "nt", "sc1", "sc0 sc1"
WPOL = os.environ.get("T32_WPOL", "")                     # weight-DMA cache policy variant
XDEF = os.environ.get("T32_DEFS", "")                      # extra -D flags of an experiment build, e.g. "T32_SETPRIO"
SO = os.path.join(REPO, "tools", "conv_t32_tools" + ("_" + POL.replace(" ", "") if POL else "") + ("_w" + WPOL.replace(" ", "") if WPOL else "") + ("_" + XDEF.replace(" ", "_") if XDEF else "") + ".bin")
SRC = os.path.join(REPO, "synergize_motion_appearance_amd", "csrc", "conv3x3_bf16_t32.hip")
if "--build" in sys.argv:
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-comment", "-mllvm", "-amdgpu-mfma-vgpr-form", "-DSMX_TOOLS", f'-DT32_RPOL="{POL}"', f'-DT32_WPOL="{WPOL}"'] + ["-D" + d for d in XDEF.split()] + [
                           "-I", os.path.join(REPO, "include"), "-I", os.path.dirname(SRC), SRC, "-o", SO])
    print("built", SO)
    sys.exit(0)
import torch  # noqa: E402
sys.path.insert(0, REPO)
from synergize_motion_appearance_amd import ops  # noqa: E402  (loads torch's HIP runtime first)

lib = C.CDLL(SO)
_p, _i = C.c_void_p, C.c_int
lib.smx_conv3x3_bf16_t32_abl.argtypes = [_p, _i, _p, _p, _p, _i, _i, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p, _i, _p, _p, _i]
B = int([a for a in sys.argv[1:] if a.isdigit()][0]) if [a for a in sys.argv[1:] if a.isdigit()] else 300
BF = torch.bfloat16
SHAPES = [(64, 64, 256), (128, 128, 128), (256, 128, 64), (512, 256, 32)]
ABL = [(0, "full"), (1, "-region loads"), (2, "-weight DMA"), (3, "-loads -DMA"), (4, "-region store"), (7, "-loads -DMA -store"), (8, "-MFMA"), (16, "-frag reads"),
       (24, "-MFMA -reads"), (32, "-epilogue traffic"), (39, "-all staging -epilogue"), (31, "epilogue only (+prologue)"), (63, "skeleton"), (64, "-epilogue"), (103, "-staging -epilogue (prologue + MFMA + reads)"), (127, "skeleton -epilogue (launch + prologue + barriers)")]


def timed(fn, n=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for gn in (0, 1):
    for cin, cout, s in SHAPES:
        x = torch.randn((B, s, s, cin), device="cuda").to(BF)
        cv = ops.Conv.from_torch(torch.randn((cout, cin, 3, 3), device="cuda") / (3 * cin ** 0.5), torch.randn(cout, device="cuda") * 0.1)
        wp = cv.w16_t32
        out = torch.empty((B, s, s, cout), device="cuda", dtype=BF)
        ss = torch.rand((B, cin, 2), device="cuda")
        fl = 2.0 * B * s * s * cout * 9 * cin
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        row = []
        only = os.environ.get("T32_ABL")
        for a, name in [x for x in ABL if only is None or str(x[0]) in only.split(",")]:
            print(f"  .. GN={gn} {cin}->{cout}@{s} abl={a}", file=sys.stderr, flush=True)
            def run():
                rc = lib.smx_conv3x3_bf16_t32_abl(x.data_ptr(), cin, wp.data_ptr(), cv.b.data_ptr(), None, 0, 0, out.data_ptr(), cout, B, s, s, cin, cout, 0, 0,
                                                  ss.data_ptr() if gn else None, gn, None, st, a)
                assert rc == 0, rc
            t = timed(run)
            row.append(f"{name}: {1e3 * t:7.1f}us {fl / t / 1e9:5.0f}TF")
        print(f"GN={gn} {cin:4d}->{cout:4d} @{s:3d} | " + " | ".join(row), flush=True)
