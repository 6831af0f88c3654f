#!/usr/bin/env python3
"""Run ONE GEMM configuration repeatedly (for rocprofv3 --pmc passes): gemm_one.py form tile M N K [reps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
os.environ["BENCH_SHAPES"] = "none"
import importlib.util
spec = importlib.util.spec_from_file_location("gb", os.path.join(os.path.dirname(os.path.abspath(__file__)), "gemm_bench.py"))
gb = importlib.util.module_from_spec(spec); spec.loader.exec_module(gb)
form, tile, M, N, K = (int(x) for x in sys.argv[1:6])
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 10
tf, ms = gb.bench(form, tile, M, N, K, reps=reps)
print(f"form={form} tile={tile} M={M} N={N} K={K} {ms*1e3:.1f} us {tf:.1f} TF/s")
