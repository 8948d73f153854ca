"""Build libsmx.so (the C-ABI HIP library, include/smx.h) for gfx950 with hipcc, in-tree.

hipcc cross-compiles without a GPU; the .so is git-ignored but travels with the tree."""
import hashlib
import json
import os
import socket
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


STAMP = os.path.join(LIBDIR, "build_stamp.json")


def _digest(paths, extra=""):
    h = hashlib.sha256(extra.encode())
    for p in paths:
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def read_stamp():
    """{"host": ..., "objects": {name: sha256 of (flags, source, headers)}} of the last build, or None."""
    try:
        with open(STAMP) as f:
            return json.load(f)
    except (OSError, ValueError):
        return None


def built_on_this_host():
    st = read_stamp()
    return bool(st) and st.get("host") == socket.gethostname() and os.path.exists(LIB)


def build(force=False, verbose=True):
    """Staleness is decided by CONTENT (sha256 of flags + source + headers against `lib/build_stamp.json`), not by mtimes: a tree
    that travelled with prebuilt objects cannot pass for fresh after an edit.  SMX_TOOLS=1 in the environment adds -DSMX_TOOLS:
    the timing-only ablation / trace instantiations (tools/wino_bench.py, tools/wino_trace.py, tools/conv16_phase.py); the
    shipped library rejects a non-zero `wino_ablate` instead."""
    os.makedirs(LIBDIR, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "hipcc")
    headers = sorted([os.path.join(REPO, "include", "smx.h")] + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")])
    flags = FLAGS + (["-DSMX_TOOLS"] if os.environ.get("SMX_TOOLS") else [])
    old = (read_stamp() or {}).get("objects", {})
    new = {}
    objs, jobs = [], []
    for src in sources():
        name = os.path.basename(src)[:-4] + ".o"
        obj = os.path.join(LIBDIR, name)
        objs.append(obj)
        new[name] = _digest([src] + headers, " ".join(a for a in flags if not a.startswith(REPO)))
        if force or not os.path.exists(obj) or old.get(name) != new[name]:
            jobs.append([hipcc] + flags + ["-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print("[smx build]", " ".join(cmd[-4:]), flush=True)
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + r.stdout)
    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    if force or jobs or not os.path.exists(LIB):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", LIB])
    if force or jobs or read_stamp() is None:
        with open(STAMP, "w") as f:
            json.dump({"host": socket.gethostname(), "objects": new}, f, indent=1)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
