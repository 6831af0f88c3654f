#!/usr/bin/env python3
"""Turn a rocprofv3 rocpd database (ROCm 7.2 default output of --kernel-trace --stats) into the
per-kernel summary table committed under profiles/.  Usage: summarize_rocpd.py results.db > summary.md"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, total_calls, total_duration, average, percentage from top_kernels order by total_duration desc").fetchall()
tot = sum(r[2] for r in rows)
print("| kernel | calls | total ms | avg us | % |")
print("|---|---:|---:|---:|---:|")
for name, calls, dur, avg, pct in rows:
    name = name.replace("mtts::", "").replace("void ", "")
    if len(name) > 90:
        name = name[:87] + "..."
    print(f"| `{name}` | {calls} | {dur / 1e3:.2f} | {avg:.1f} | {pct:.2f} |")
print(f"\ntotal kernel time {tot / 1e3:.1f} ms over {sum(r[1] for r in rows)} dispatches")
