#!/usr/bin/env python3
"""Audit of the asm-issued region requests of csrc/conv3x3_bf16_t32.hip in the generated gfx950 ISA (no GPU needed: hipcc -S).

The K loop loads its region chunks with inline-asm `global_load_dwordx4` (tagged `t32-request`) that hipcc does not count; the data is valid
only behind the matching `s_waitcnt vmcnt(N) ; t32-claim v[a:b]` statement, one step later.  Between the two the compiler must not read,
copy, spill or overwrite those registers (it is allowed to: it believes they were written at the request).  This walks every
conv3x3_t32_kernel instantiation in text order and reports any instruction that names a register with a request in flight.
usage: python tools/t32_isa_audit.py [file.s]   (without an argument: compiles the kernel to a temporary .s)"""
import os
import re
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(REPO, "synergize_motion_appearance_amd", "csrc", "conv3x3_bf16_t32.hip")
sys.path.insert(0, REPO)
from synergize_motion_appearance_amd.build import flags_for  # noqa: E402  (the library's own flags for this source, minus -fPIC)
FLAGS = [f for f in flags_for(SRC) if f != "-fPIC"]
VREG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def regs_of(text):
    out = set()
    for m in VREG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def compile_isa(extra=()):
    out = os.path.join(tempfile.mkdtemp(prefix="t32isa"), "t32.s")
    subprocess.check_call(["hipcc"] + FLAGS + list(extra) + ["-S", "--cuda-device-only", SRC, "-o", out], stderr=subprocess.DEVNULL)
    return out


def _kernels(path):
    cur, body = None, []
    for n, line in enumerate(open(path), 1):
        t = line.strip()
        m = re.match(r"^(_ZN\S*conv3x3_t32_kernelILi\d+E\S*):", t)
        if m:
            cur, body = m.group(1), []
            continue
        if cur is not None:
            if t.startswith(".Lfunc_end"):
                yield cur, body
                cur = None
            else:
                body.append((n, t))


def _walk(lines, pending, flag, bad, kern):
    """one in-order pass over straight-line code; `pending`: register -> line of its request"""
    rq = cl = 0
    for n, t in lines:
        if not t or t.startswith((";", ".", "//")):
            continue
        code = t.split(";")[0]
        if "t32-request" in t:
            for r in regs_of(code.split(",")[0]):
                pending[r] = n
            rq += 1
        elif "t32-claim" in t:
            for r in regs_of(t.split("t32-claim")[1]):
                pending.pop(r, None)
            cl += 1
        elif flag:
            hit = regs_of(code) & set(pending)
            if hit:
                bad.append((kern, n, t, sorted(hit)))
    return rq, cl


def audit(path):
    """-> (kernels seen, requests, claims, loops with requests, violations [(kernel, line number, text, registers)]).
    Two checks per kernel instantiation: (1) every innermost loop that issues requests (the K loops: straight-line bodies) is walked
    TWICE in order -- the second pass starts with what the first left in flight, exactly the steady state -- and any instruction naming
    a register in flight is a violation; (2) outside loops, in text order (fall-through), nothing touches a register between its request
    and its claim, the next unconditional branch, or the entry of a K loop."""
    kernels = requests = claims = loops = 0
    bad = []
    for kern, body in _kernels(path):
        kernels += 1
        idx = {t.split(":")[0]: i for i, (n, t) in enumerate(body) if re.match(r"^\.LBB\d+_\d+:", t)}
        i = 0
        in_loop = [False] * len(body)
        for i, (n, t) in enumerate(body):
            m = re.match(r"^s_cbranch_\w+\s+(\.LBB\d+_\d+)", t) or re.match(r"^s_branch\s+(\.LBB\d+_\d+)", t)
            if m and m.group(1) in idx and idx[m.group(1)] < i:                       # backward branch: [header, i] is a loop
                lo = idx[m.group(1)]
                seg = body[lo:i + 1]
                inner = not any(re.match(r"^\.LBB\d+_\d+:", t2) and k > 0 and "Loop Header" in t2 for k, (n2, t2) in enumerate(seg))
                if inner and any("t32-request" in t2 for _, t2 in seg):
                    loops += 1
                    pend = {}
                    rq, cl = _walk(seg, pend, False, bad, kern)
                    _walk(seg, pend, True, bad, kern)
                    requests += rq
                    claims += cl
                    for k in range(lo, i + 1):
                        in_loop[k] = True
        pend = {}
        for k, (n, t) in enumerate(body):
            if in_loop[k]:
                pend = {}
                continue
            if re.match(r"^\.LBB\d+_\d+:", t) or t.startswith(("s_cbranch", "s_branch", "s_endpgm")):
                pend = {}
                continue
            rq, cl = _walk([(n, t)], pend, True, bad, kern)
            requests += rq
            claims += cl
    return kernels, requests, claims, loops, bad


if __name__ == "__main__":
    path = sys.argv[1] if len(sys.argv) > 1 else compile_isa()
    k, rq, cl, lp, bad = audit(path)
    print(f"{path}: {k} kernel(s), {rq} requests, {cl} claims, {lp} K loop(s) walked twice, {len(bad)} violation(s)")
    for kern, n, t, regs in bad[:40]:
        print(f"  line {n}: {t}   <- in flight: v{regs}")
    sys.exit(1 if bad or not k or not rq or not lp else 0)
