#!/bin/bash
# round 4, call b: slices-in-flight sweep of the bf16 K-loop on under-filled shapes, per-launch GEMM dump of C2 (both numerics modes),
# single-task-rank baselines of this build.   -> gpurun_out/r04b/
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r04b; mkdir -p $OUT; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build_device()" > $OUT/build.log 2>&1
BENCH_BF16=1 BENCH_TILES=1064,2064,4064,1128,2128,3128 BENCH_SHAPES="conv1_fwd,conv1_dgrad,conv2_fwd,1task conv1,1task fc,1task dgrad,dec 1 task" timeout 300 python tools/gemm_bench.py > $OUT/gemm_bf16_pf.txt 2>&1; cat $OUT/gemm_bf16_pf.txt
BENCH_BF16=1 BENCH_TILES=1064,4064,2128 BENCH_CUSTOM="c2 conv1,6900,1024,2304;c2 dgrad,6900,256,9216;c2 fc,6900,256,256;c2 conv2,6900,256,1024;c2 qkv,6900,768,256" timeout 300 python tools/gemm_bench.py > $OUT/gemm_bf16_c2.txt 2>&1; cat $OUT/gemm_bf16_c2.txt
BENCH_TILES=3064,4064 BENCH_CUSTOM="c2 conv1,6900,1024,2304;c2 dgrad,6900,256,9216;c2 fc,6900,256,256;c2 conv2,6900,256,1024;c2 qkv,6900,768,256" timeout 300 python tools/gemm_bench.py > $OUT/gemm_f32_c2.txt 2>&1; cat $OUT/gemm_f32_c2.txt
MTTS_GEMM_DUMP=$OUT/c2_dump.csv C2_ITERS=6 timeout 300 python tools/c2_bench.py > $OUT/c2.json 2> $OUT/c2.err; python tools/gemm_sites.py $OUT/c2_dump.csv > $OUT/c2_sites_bf16.md 2>&1; head -45 $OUT/c2_sites_bf16.md
X="--no-cpu-baseline --no-inference --no-frontend --no-baseline-c2"
timeout 400 python bench.py --steps 10 --warmup 3 --emulate-world 8 $X > $OUT/bench_w8.json 2> $OUT/bench_w8.err
python - <<'PY'
import json
j = json.load(open("gpurun_out/r04b/bench_w8.json"))
print("w8 FO", j["ms_per_step"], "SO", j.get("second_order", {}).get("ms_per_step"))
PY
