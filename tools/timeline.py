#!/usr/bin/env python3
"""Timeline of a rocprofv3 --kernel-trace rocpd database: per-queue busy time, union busy time, idle gaps and the kernels that
occupy the longest queue — what a latency-bound step (single-task rank) is waiting on.
CAUTION for latency-bound steps: under rocprofv3 the host pays ~2x per launch — a single-task rank's step is 37 ms instead of 30 and the host, not the
device, sets its pace (bench line: host_enqueue_ms_from_idle 34.9 ms under the profiler, 17.2 ms without) — so the waits this prints in front of the
kernels that follow a burst of side-stream launches are the profiled host's, not the product's.
Usage: timeline.py results.db [skip_fraction]   (the first skip_fraction of the trace — warm-up, engine creation — is ignored)"""
import collections
import sqlite3
import sys


def union(iv):
    iv = sorted(iv)
    busy, (cs, ce) = 0, iv[0]
    gaps = []
    for s, e in iv[1:]:
        if s > ce:
            busy += ce - cs
            gaps.append(s - ce)
            cs, ce = s, e
        else:
            ce = max(ce, e)
    return busy + ce - cs, gaps


def main():
    c = sqlite3.connect(sys.argv[1])
    skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
    rows = c.execute("select name, start, end, queue_id, stream_id, grid_x, grid_y, grid_z, workgroup_x from kernels order by start").fetchall()
    t0, t1 = rows[0][1], max(r[2] for r in rows)
    cut = t0 + skip * (t1 - t0)
    rows = [r for r in rows if r[1] >= cut]
    span = (max(r[2] for r in rows) - rows[0][1]) / 1e6
    busy, gaps = union([(r[1], r[2]) for r in rows])
    print(f"window {span:.2f} ms, {len(rows)} dispatches, union busy {busy / 1e6:.2f} ms ({100 * busy / 1e6 / span:.1f} %), "
          f"{len(gaps)} idle gaps: total {sum(gaps) / 1e6:.2f} ms, median {sorted(gaps)[len(gaps) // 2] / 1e3 if gaps else 0:.2f} us")
    perq = collections.defaultdict(list)
    for r in rows:
        perq[(r[3], r[4])].append(r)
    print("| queue, stream | dispatches | busy ms | % of window |")
    print("|---|---:|---:|---:|")
    for k, v in sorted(perq.items(), key=lambda kv: -sum(r[2] - r[1] for r in kv[1])):
        b = sum(r[2] - r[1] for r in v) / 1e6
        print(f"| {k} | {len(v)} | {b:.2f} | {100 * b / span:.1f} |")
    main_q = max(perq.items(), key=lambda kv: sum(r[2] - r[1] for r in kv[1]))[1]
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in main_q:
        n = r[0].replace("mtts::", "").replace("void ", "")
        n = n[:70]
        agg[n][0] += 1
        agg[n][1] += (r[2] - r[1]) / 1e6
    print("\nbusiest queue, by kernel:")
    print("| kernel | calls | total ms | avg us |")
    print("|---|---:|---:|---:|")
    for n, (k, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
        print(f"| `{n}` | {k} | {ms:.2f} | {1e3 * ms / k:.1f} |")
    # gaps INSIDE the busiest queue (the critical stream): idle time in front of each dispatch (end of the previous dispatch of the same queue ->
    # its start), gaps above 200 us (step boundaries, host work) left out; attributed to the kernel that starts after the gap
    mq = sorted(main_q, key=lambda r: r[1])
    gap_by = collections.defaultdict(lambda: [0, 0.0])
    gtot, gn, ghist = 0.0, 0, collections.Counter()
    for a, b in zip(mq, mq[1:]):
        g = (b[1] - a[2]) / 1e3
        if 0 < g < 200:
            n = b[0].replace("mtts::", "").replace("void ", "")[:50]
            gap_by[n][0] += 1; gap_by[n][1] += g
            gtot += g; gn += 1
            ghist[min(int(g // 2) * 2, 20)] += 1
    print(f"\nbusiest queue: {gn} gaps below 200 us between consecutive dispatches, total {gtot / 1e3:.2f} ms, mean {gtot / max(gn, 1):.2f} us; "
          "histogram (us bucket: count) " + ", ".join(f"{k}{'+' if k == 20 else ''}: {v}" for k, v in sorted(ghist.items())))
    pair = collections.defaultdict(lambda: [0, 0.0])
    for a, b in zip(mq, mq[1:]):
        g = (b[1] - a[2]) / 1e3
        if 15 <= g < 2000:
            k = (a[0].replace("mtts::", "").replace("void ", "").split("(")[0][:40], b[0].replace("mtts::", "").replace("void ", "").split("(")[0][:40])
            pair[k][0] += 1; pair[k][1] += g
    print("\nbusiest queue: waits of 15 us .. 2 ms, by (kernel before -> kernel after):")
    print("| before -> after | waits | total ms | mean us |")
    print("|---|---:|---:|---:|")
    for k, (n_, us) in sorted(pair.items(), key=lambda kv: -kv[1][1])[:16]:
        print(f"| `{k[0]}` -> `{k[1]}` | {n_} | {us / 1e3:.2f} | {us / n_:.1f} |")
    print()
    print("| kernel that starts after the gap | gaps | total ms | mean us |")
    print("|---|---:|---:|---:|")
    for n, (k, us) in sorted(gap_by.items(), key=lambda kv: -kv[1][1])[:14]:
        print(f"| `{n}` | {k} | {us / 1e3:.2f} | {us / k:.1f} |")
    # duration histogram of the busiest queue's kernels
    edges = [5, 10, 20, 40, 80, 160, 320, 1e9]
    hist = [[0, 0.0] for _ in edges]
    for r in main_q:
        us = (r[2] - r[1]) / 1e3
        for i, e in enumerate(edges):
            if us < e:
                hist[i][0] += 1; hist[i][1] += us / 1e3
                break
    print("\nbusiest queue, kernel duration histogram (us: count, total ms): " +
          ", ".join(f"<{int(e) if e < 1e9 else 'inf'}: {h[0]}, {h[1]:.2f}" for e, h in zip(edges, hist)))


if __name__ == "__main__":
    main()
