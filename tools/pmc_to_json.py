#!/usr/bin/env python3
"""Fold rocprofv3 --pmc rocpd databases (one pass per counter group, as the MI355X guide prescribes) into the per-kernel
JSON that bench.py's roofline leg reads (profiles/rNN_pmc_hbm.json).

usage: pmc_to_json.py OUT.json[:KEY] DB [DB ...]      (KEY = top-level key of the per-kernel table, default "kernels"; an existing
OUT.json is updated, so the first- and second-order passes can share one file)
Counters used when present: FETCH_SIZE, WRITE_SIZE (KB; gfx950: FETCH_SIZE counts 64 B per 128-B request for 16-B/lane
loads -> doubled), SQ_BUSY_CYCLES, SQ_VALU_MFMA_BUSY_CYCLES (MFMA-pipe busy fraction = MFMA_BUSY / (BUSY * 32 SIMDs per SE),
both summed over the shader engines)."""
import collections
import json
import re
import sqlite3
import sys

def categories(sym):
    """Keys a dispatch is accumulated under: the kernel's name as rocprofv3 --stats prints it (template arguments included, what
    bench.py's roofline.kernel holds) and the bare template name (all instantiations together)."""
    m = re.search(r"(gemm_\w+_kernel)(<[^>(]*>)?", sym)
    if not m:
        return []
    return [m.group(1) + (m.group(2) or ""), m.group(1)] if m.group(2) else [m.group(1)]


def name_column(c):
    cols = [r[1] for r in c.execute("pragma table_info(rocpd_info_kernel_symbol)")]
    for want in ("demangled_kernel_name", "display_name", "formatted_kernel_name", "kernel_name"):
        if want in cols:
            return want
    return "kernel_name"


acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))  # cat -> counter -> [sum, dispatches]
for path in sys.argv[2:]:
    c = sqlite3.connect(path)
    rows = c.execute(f"""select s.{name_column(c)}, p.name, d.id, sum(e.value) from rocpd_pmc_event e
        join rocpd_info_pmc p on e.pmc_id = p.id
        join rocpd_kernel_dispatch d on e.event_id = d.event_id
        join rocpd_info_kernel_symbol s on d.kernel_id = s.id group by 1, 2, 3""").fetchall()
    for sym, ctr, _disp, val in rows:
        for cat in categories(sym):
            a = acc[cat][ctr]
            a[0] += val
            a[1] += 1
out = {"source": "rocprofv3 --pmc passes (one counter group per pass: FETCH_SIZE | WRITE_SIZE | SQ_BUSY_CYCLES + SQ_VALU_MFMA_BUSY_CYCLES) over `bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-inference --no-frontend --no-baseline-c2 --no-roofline` (fp32, dropout on; `--order 2` for kernels_second_order), 1x MI355X",
       "correction": "gfx950: FETCH_SIZE counts 64 B per 128-B request for 16-B/lane loads -> doubled (MI355X_MICROARCH.md, HBM section); WRITE_SIZE as reported; both KB",
       "kernels": {}}
for cat, d in sorted(acc.items()):
    k = {}
    if "FETCH_SIZE" in d:
        k["fetch_size_kb_per_launch_raw"] = round(d["FETCH_SIZE"][0] / d["FETCH_SIZE"][1], 1)
        k["launches_sampled"] = d["FETCH_SIZE"][1]
    if "WRITE_SIZE" in d:
        k["write_size_kb_per_launch_raw"] = round(d["WRITE_SIZE"][0] / d["WRITE_SIZE"][1], 1)
    if "fetch_size_kb_per_launch_raw" in k and "write_size_kb_per_launch_raw" in k:
        k["hbm_bytes_per_launch"] = int(1024 * (2 * k["fetch_size_kb_per_launch_raw"] + k["write_size_kb_per_launch_raw"]))
    if "SQ_BUSY_CYCLES" in d and "SQ_VALU_MFMA_BUSY_CYCLES" in d and d["SQ_BUSY_CYCLES"][0] > 0:
        k["mfma_busy_frac"] = round(d["SQ_VALU_MFMA_BUSY_CYCLES"][0] / (32.0 * d["SQ_BUSY_CYCLES"][0]), 4)
    out["kernels"][cat] = k
import os
target, _, key = sys.argv[1].partition(":")
key = key or "kernels"
if key != "kernels":
    out[key] = out.pop("kernels")
if os.path.exists(target):
    old = json.load(open(target))
    old.update({k: v for k, v in out.items() if k not in ("source", "correction") or k not in old})
    out = old
json.dump(out, open(target, "w"), indent=1)
print(json.dumps(out[key], indent=1))
