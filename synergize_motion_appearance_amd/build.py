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
# No packed fp32 instructions (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) in the library's generated code.  Measured on the MI355X (DESIGN
# section 6, profiles/r05_shared_device_*.txt): `sparse_motion_kernel`, whose keypoint arithmetic hipcc SLP-packs into such instructions, returned wrong
# values in lanes 48-63 of some waves whenever ANOTHER process on the same GPU was running the bf16 MFMA convolution -- identical inputs, a
# different output, 10-30 % of the frames of a shared-device run; rebuilt without the packed forms the same kernel is bit-stable
# (tools/sm_probe.hip, 0 of 346,592 launches against 187 M wrong elements), and so is the whole pipeline (0 of 9,269 passes against 50 of
# 1,883).  Every other kernel was checked as the victim of the same neighbour and is clean either way; the two Winograd files keep the
# packed forms (their input transform is where they pay: -2.4 % on the fp32 headline without them).  The flag is a device-side target
# feature; the host half of the compilation prints a harmless "not a recognized feature" note.
NO_PACKED_FP32 = ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]
KEEPS_PACKED_FP32 = ("winograd.hip", "winograd43.hip")


def flags_for(src):
    return FLAGS + ([] if os.path.basename(src) in KEEPS_PACKED_FP32 else NO_PACKED_FP32)


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
    extra = (["-DSMX_TOOLS"] if os.environ.get("SMX_TOOLS") else []) + os.environ.get("SMX_HIPCC_EXTRA", "").split()   # EXTRA: A/B builds (tools)
    old = (read_stamp() or {}).get("objects", {})
    new = {}
    objs, jobs = [], []
    for src in sources():
        name = os.path.basename(src)[:-4] + ".o"
        obj = os.path.join(LIBDIR, name)
        objs.append(obj)
        flags = flags_for(src) + extra
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
