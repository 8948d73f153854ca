#!/usr/bin/env python3
"""Turn a rocprofv3 (--kernel-trace --stats) rocpd SQLite database into a text kernel
summary that can be committed under profiles/.   usage: rocprof_summary.py results.db out.txt"""
import sqlite3
import sys


def main(db_path, out_path):
    db = sqlite3.connect(db_path)
    rows = db.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    with open(out_path, "w") as f:
        f.write(f"# rocprofv3 --kernel-trace --stats ; source: {db_path}\n")
        f.write("# durations in microseconds\n")
        f.write(f"{'calls':>7} {'total_us':>12} {'avg_us':>10} {'pct':>6}  kernel\n")
        for name, calls, tot, avg, pct in rows:
            name = name.replace("(anonymous namespace)::", "")
            if len(name) > 150:
                name = name[:147] + "..."
            f.write(f"{calls:7d} {tot:12.1f} {avg:10.2f} {pct:6.2f}  {name}\n")
    print(f"wrote {out_path} ({len(rows)} kernels)")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
