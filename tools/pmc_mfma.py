#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc pass (counter_collection.csv) of `bench.py` into per-kernel-family MFMA utilisation.
usage: pmc_mfma.py <pmc_dir> <out.json> [command string]

Counters expected in ONE pass (SQ has 8 slots, GRBM 2; see MI355X_MICROARCH.md "rocprofv3 PMC slots"):
  SQ_INSTS_VALU_MFMA_MOPS_F32 (or _BF16 / _F16)  matrix ops in units of 512 flops (a v_mfma_f32_32x32x2_f32 = 4096 flops = 8)
  SQ_VALU_MFMA_BUSY_CYCLES                 cycles the matrix pipe is busy, summed over the chip's 1024 SIMDs (64 per 32x32x2 f32 MFMA)
  SQ_INSTS_MFMA, SQ_BUSY_CYCLES, SQ_WAVE_CYCLES (optional)
  GRBM_GUI_ACTIVE                          active shader-clock cycles summed over the 8 XCDs
Derived per family:  clock = GUI_ACTIVE/8/duration;  mfma_busy = MFMA_BUSY / (GUI_ACTIVE/8 * 1024 SIMDs) (fraction of the
cycles the chip actually ran);  executed TFLOP/s = MOPS*512/duration;  frac_of_peak = executed / dense peak (= mfma_busy * clock/2.4 GHz).
Durations are the dispatch timestamps of the SAME (counter-collecting, hence serialised and slower-clocked) pass."""
import collections
import csv
import glob
import json
import sys

SIMDS, XCDS, SPEC_GHZ = 1024, 8, 2.4
PEAK = {"F32": 157.3, "BF16": 2500.0, "F16": 2500.0}
FAMILIES = (("winograd_bf3_kernel", "winograd_bf3"), ("winograd_wide_kernel", "winograd"), ("winograd_kernel", "winograd"), ("gemm_conv_kernel", "gemm_conv"), ("gemm_bf16_kernel", "gemm_bf16"), ("gemm_rp_bf16_kernel", "gemm_bf16"), ("gemm_rp_f32_kernel", "gemm_conv"), ("gemm_rp_bf3_kernel", "gemm_bf3"), ("attn_bf3_kernel", "attention_bf3"), ("attn_f16_kernel", "attention_bf3"), ("conv7_bf16x3", "conv7_x3"), ("attnblock16", "attnblock"),
            ("conv3x3_bf16_kernel", "conv3x3_bf16"), ("conv3x3_t32_kernel", "conv3x3_bf16"), ("attn_mfma16", "attention_mfma16"), ("attn_mfma", "attention_mfma"), ("vq_kernel", "vq"))


def family(name):
    for key, fam in FAMILIES:
        if key in name:
            return fam
    return None


def variants(name):
    """family + finer keys: the Winograd block shapes separately (the 8-wave N=64 block is the big-launch variant)."""
    fam = family(name)
    if not fam:
        return []
    out = [fam]
    if fam == "winograd":
        out.append("winograd_wide" if "winograd_wide_kernel" in name else
                   "winograd_nw2" if "winograd_kernel<false, 2" in name or "winograd_kernel<true, 2" in name else "winograd_nw1")
    return out


def _library_build():
    """{object: sha256 of (flags + source + headers)} of the libsmx.so these counters were taken on (lib/build_stamp.json): bench.py quotes a
    family's counters only while the objects that hold its kernels still have these digests."""
    import os
    p = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "synergize_motion_appearance_amd", "lib", "build_stamp.json")
    try:
        return json.load(open(p))["objects"]
    except (OSError, ValueError, KeyError):
        return None


def main(pmc_dir, out, command=""):
    per = collections.defaultdict(lambda: collections.defaultdict(float))   # family -> counter -> sum
    disp = collections.defaultdict(dict)                                   # family -> dispatch id -> duration ns
    for f in glob.glob(pmc_dir + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            for fam in variants(r["Kernel_Name"]):
                per[fam][r["Counter_Name"]] += float(r["Counter_Value"])
                disp[fam][r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    res = {}
    for fam, c in per.items():
        n = len(disp[fam])
        dur_s = sum(disp[fam].values()) * 1e-9
        e = {"launches_profiled": n, "avg_launch_us": round(1e6 * dur_s / n, 2)}
        gui = c.get("GRBM_GUI_ACTIVE", 0.0)
        if gui:
            e["shader_clock_GHz"] = round(gui / XCDS / dur_s / 1e9, 3)
        for dt in ("F32", "BF16", "F16"):
            mops = c.get(f"SQ_INSTS_VALU_MFMA_MOPS_{dt}", 0.0)
            if mops:
                tf = mops * 512.0 / dur_s / 1e12
                e[f"executed_TFLOPs_{dt.lower()}"] = round(tf, 2)
                e[f"frac_of_{dt.lower()}_mfma_peak"] = round(tf / PEAK[dt], 4)
        if c.get("SQ_VALU_MFMA_BUSY_CYCLES") and gui:
            e["mfma_busy_frac_of_active_cycles"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (gui / XCDS * SIMDS), 4)
        if c.get("SQ_INSTS_MFMA"):
            e["mfma_insts_per_launch"] = round(c["SQ_INSTS_MFMA"] / n)
            if c.get("SQ_VALU_MFMA_BUSY_CYCLES"):
                e["busy_cycles_per_mfma"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / c["SQ_INSTS_MFMA"], 2)
        if c.get("SQ_BUSY_CYCLES") and gui:
            e["sq_busy_frac"] = round(c["SQ_BUSY_CYCLES"] / gui, 4) if c["SQ_BUSY_CYCLES"] <= gui * 1.01 else None
        for k in ("SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY", "SQ_LDS_BANK_CONFLICT", "SQ_INSTS_VALU", "SQ_INSTS_LDS"):
            if k in c:
                e[k + "_per_launch"] = round(c[k] / n)
        res[fam] = e
    json.dump({"source": "rocprofv3 --pmc (one pass) -- " + command, "definitions": __doc__.split("Derived per family:")[1].strip(),
               "kernels": res, "library_build": _library_build()}, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], " ".join(sys.argv[3:]))
