"""Build libsmx.so (the C-ABI HIP library, include/smx.h) for gfx950 with hipcc, in-tree.

hipcc cross-compiles without a GPU; the .so is git-ignored but travels with the tree."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIBDIR = os.path.join(PKG, "lib")
LIB = os.path.join(LIBDIR, "libsmx.so")
# -amdgpu-mfma-vgpr-form: MFMA results go to architectural VGPRs (gfx90a+ unified file) instead of AccVGPRs, so the VALU code
# that consumes accumulators (softmax on S, O rescale, Winograd / GEMM epilogues) needs no v_accvgpr_read/write copies
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-comment", "-mllvm", "-amdgpu-mfma-vgpr-form",
         "-I", os.path.join(REPO, "include"), "-I", CSRC]


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    """SMX_TOOLS=1 in the environment adds -DSMX_TOOLS: the timing-only ablation / trace instantiations of the Winograd kernels
    (tools/wino_bench.py, tools/wino_trace.py); the shipped library rejects a non-zero `wino_ablate` instead."""
    os.makedirs(LIBDIR, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "hipcc")
    headers = [os.path.join(REPO, "include", "smx.h")] + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    objs, jobs = [], []
    for src in sources():
        obj = os.path.join(LIBDIR, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if force or _stale(obj, [src] + headers):
            jobs.append([hipcc] + FLAGS + (["-DSMX_TOOLS"] if os.environ.get("SMX_TOOLS") else []) + ["-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print("[smx build]", " ".join(cmd[-4:]), flush=True)
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + r.stdout)
    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    if force or jobs or _stale(LIB, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", LIB])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
