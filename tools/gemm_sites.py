#!/usr/bin/env python3
"""Aggregate the per-launch GEMM CSV written by the profiler (MTTS_GEMM_DUMP=<path> python bench.py ...) by call-site
shape class (form, tile, N, K, row-count bucket): launches, total us, achieved TFLOP/s, share of the GEMM time."""
import collections
import csv
import sys

FORMS = ["NT", "NN", "TN", "multi", "attn"]
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    key = (FORMS[int(r["form"])] + "/k" + r.get("kind", "?") + ("" if r.get("ctx", "0") == "0" else "/s" + r["ctx"]), int(r["tile"]), int(r["N"]), int(r["K"]), int(float(r["rows"])), int(r["groups"]), int(r["splitk"]))
    a = agg[key]
    a[0] += 1; a[1] += float(r["us"]); a[2] += float(r["gflop"])
tot = sum(a[1] for a in agg.values())
print(f"total GEMM time {tot / 1e3:.2f} ms, {sum(a[2] for a in agg.values()) / 1e3:.3f} TFLOP")
print("(form/kN: N = kernel kind index of csrc/gemm.h GemmKind — 3-5 gemm_f32_kernel<F,64,64,32>, 0-2 <F,64,64,16>, 9-11 gemm_glds_kernel<F>, 12 / 13 gemm_f32_multi_kernel<64,64,16 / 32>, 14 gemm_glds_multi_kernel, 18-20 the dual-source kernels; multi: N = problems in the launch, K = its longest K-loop, groups = workgroups; /s1, /s2 = launched on the weight-gradient / run-ahead side stream: timed on that stream, UNDER the main stream's launches)")
print("| form | tile | N | K | rows | groups | splitK | launches | total us | avg us | TFLOP/s | % |")
print("|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|")
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"| {k[0]} | {k[1]} | {k[2]} | {k[3]} | {k[4]} | {k[5]} | {k[6]} | {a[0]} | {a[1]:.0f} | {a[1] / a[0]:.1f} | {a[2] / a[1] * 1e3 if a[1] else 0:.1f} | {100 * a[1] / tot:.1f} |")
