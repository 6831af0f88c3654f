import os, sys
sys.path.insert(0, "/root/repo/tools"); sys.path.insert(0, "/root/repo")
os.environ["BENCH_SHAPES"] = "none"
import importlib.util
spec = importlib.util.spec_from_file_location("gb", "/root/repo/tools/gemm_bench.py"); gb = importlib.util.module_from_spec(spec); spec.loader.exec_module(gb)
names = {0: "full", 1: "-gload", 2: "-lds_store", 4: "-frag_read", 8: "-barrier", 3: "-gload-store", 6: "-store-read", 7: "-gload-store-read", 14: "-store-read-barrier", 15: "MFMA only"}
for name, M, N, K in (("lone WG/CU: 132 WGs", 2100, 256, 1024), ("2 WG/CU: 528 WGs", 2100, 1024, 1024), ("full: 4272 WGs", 17047, 1024, 2304)):
    for a in (0, 1, 2, 4, 8, 3, 6, 7, 14, 15):
        tf, ms = gb.bench(0, a * 10000 + 1064, M, N, K, reps=8)
        print(f"{name:22s} {names[a]:22s} {ms*1e3:9.1f} us  {tf:6.1f} TF/s", flush=True)
