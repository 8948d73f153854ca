#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `bench.py` (separate runs, csv) into
per-kernel-family HBM-side bytes per launch.   usage: pmc_traffic.py <fetch_dir> <write_dir> <out.json>
gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE tallies the 128-B requests of wide coalesced
(16 B/lane) reads at 64 B -> doubled here for the kernels whose global reads are all float4; WRITE_SIZE
matched the algorithmic bytes of the warp / conv kernels exactly (profiles/r01_c_pmc_*) and is taken as is.
Counter values are KB."""
import collections
import csv
import glob
import json
import sys


def load(d, counter):
    agg = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return agg


def variant(name):
    """finer key beside the family: the Winograd block shapes separately -- the 4-wave 'wide' block is what the full-batch (B = 300 by default) launches of the
    big layers run, `winograd_kernel<1>` the B=1 source-encoder / small launches (same split as tools/pmc_mfma.py)."""
    if "winograd_wide_kernel" in name:
        return "winograd_wide"
    if "winograd_kernel" in name:
        return "winograd_nw1" if ("<1" in name or "<false, 1" in name) else "winograd_nw2"
    return None


def family(name):
    for key, fam in (("winograd_bf3_kernel", "winograd_bf3"), ("winograd_", "winograd"), ("gemm_conv_kernel", "gemm_conv"), ("gemm_bf16_kernel", "gemm_bf16"), ("gemm_rp_bf16_kernel", "gemm_bf16"), ("gemm_rp_f32_kernel", "gemm_conv"), ("gemm_rp_bf3_kernel", "gemm_bf3"), ("conv7_bf16x3", "conv7_x3"), ("attnblock16", "attnblock"), ("conv3x3_bf16", "conv3x3_bf16"), ("conv3x3_t32", "conv3x3_bf16"), ("attn_", "attention"),
                     ("warp_", "warp"), ("gn_", "groupnorm"), ("layernorm", "layernorm")):
        if key in name:
            return fam
    return None


def _library_build():
    """{object: sha256 of (flags + source + headers)} of the libsmx.so these counters were taken on (lib/build_stamp.json): bench.py quotes a
    family's counters only while the objects that hold its kernels still have these digests."""
    import os
    p = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "synergize_motion_appearance_amd", "lib", "build_stamp.json")
    try:
        return json.load(open(p))["objects"]
    except (OSError, ValueError, KeyError):
        return None


def main(fetch_dir, write_dir, out):
    fe, wr = load(fetch_dir, "FETCH_SIZE"), load(write_dir, "WRITE_SIZE")
    fams = collections.defaultdict(lambda: {"launches": 0, "fetch_kb_raw": 0.0, "write_kb": 0.0, "wlaunches": 0})
    for name, vals in fe.items():
        for fam in (family(name), variant(name)):
            if fam:
                fams[fam]["launches"] += len(vals)
                fams[fam]["fetch_kb_raw"] += sum(vals)
    for name, vals in wr.items():
        for fam in (family(name), variant(name)):
            if fam:
                fams[fam]["write_kb"] += sum(vals)
                fams[fam]["wlaunches"] += len(vals)
    res = {}
    for fam, v in fams.items():
        n = max(v["launches"], 1)
        res[fam] = {"launches_profiled": v["launches"], "fetch_bytes_per_launch_corrected": 2.0 * 1024 * v["fetch_kb_raw"] / n,
                    "fetch_bytes_per_launch_raw": 1024 * v["fetch_kb_raw"] / n, "write_bytes_per_launch": 1024 * v["write_kb"] / max(v["wlaunches"], 1)}
        res[fam]["hbm_bytes_per_launch"] = res[fam]["fetch_bytes_per_launch_corrected"] + res[fam]["write_bytes_per_launch"]
    conv = {k: res[k] for k in ("winograd", "winograd_bf3", "gemm_conv", "gemm_bf3") if k in res}
    nl = sum(v["launches_profiled"] for v in conv.values())
    if nl:
        res["conv_gemm_family"] = {"launches_profiled": nl,
                                   "hbm_bytes_per_launch": sum(v["hbm_bytes_per_launch"] * v["launches_profiled"] for v in conv.values()) / nl}
    json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on `python bench.py --steps 2 --warmup 1 "
                         "--profile-only` (timed steps only: no B=1 re-render check, so every warp / attention / layernorm launch is a full-batch (B = 300 by default) "
                         "launch; the B=1 source-encoder convolutions are separated by block shape: winograd_wide = the full-batch (B = 300 by default) launches); "
                         "FETCH_SIZE doubled (gfx950 128-B requests tallied at 64 B)",
               "families": res, "library_build": _library_build()}, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:4])
