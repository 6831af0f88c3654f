#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc rocpd databases: per kernel, mean of each counter over its dispatches."""
import sqlite3, sys, collections
for path in sys.argv[1:]:
    c = sqlite3.connect(path)
    try:
        rows = c.execute("""select s.kernel_name, p.name, avg(e.value), count(*) from rocpd_pmc_event e
            join rocpd_info_pmc p on e.pmc_id = p.id
            join rocpd_kernel_dispatch d on e.event_id = d.event_id
            join rocpd_info_kernel_symbol s on d.kernel_id = s.id group by 1, 2""").fetchall()
    except Exception as ex:
        print(path, "ERR", ex); continue
    by = collections.defaultdict(dict)
    for k, n, v, cnt in rows:
        by[k][n] = (v, cnt)
    for k, d in by.items():
        if "gemm" not in k and len(sys.argv) < 3: pass
        print(path.split("/")[-1], k[:70])
        for n, (v, cnt) in sorted(d.items()):
            print(f"    {n:32s} {v:16.1f}  (n={cnt})")
