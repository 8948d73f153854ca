#!/usr/bin/env python3
"""bench.py with tuning knobs set first (same-box A/B runs): python tools/bench_tuned.py name=value [name=value ...] -- <bench.py arguments>"""
import os
import runpy
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from synergize_motion_appearance_amd import ops  # noqa: E402

i = sys.argv.index("--")
for kv in sys.argv[1:i]:
    k, v = kv.split("=")
    ops.set_tuning(k, int(v))
sys.argv = [os.path.join(REPO, "bench.py")] + sys.argv[i + 1:]
runpy.run_path(sys.argv[0], run_name="__main__")
