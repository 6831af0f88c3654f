#!/bin/bash
# round 4, first GPU call: the bf16 operand family (kernel tests, micro-benchmark, C2 in both numerics modes), the kernel tests touched by the
# work-queue removal, and one default bench line (concurrent CPU baseline + gradient parity).   -> gpurun_out/r04a/
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r04a; mkdir -p $OUT; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build_device()" > $OUT/build.log 2>&1
timeout 900 python -m pytest tests/test_bf16_mode.py tests/test_gpu_kernels.py tests/test_kernel_entries.py -m gpu -q -x -s > $OUT/pytest_a.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_a.log; tail -5 $OUT/pytest_a.log
BENCH_BF16=1 BENCH_SHAPES="square4096,conv1_fwd,conv1_dgrad,conv2_fwd,qkv,postnet_mid,1task conv1,1task fc" timeout 300 python tools/gemm_bench.py > $OUT/gemm_bf16.txt 2>&1; tail -50 $OUT/gemm_bf16.txt
timeout 300 python tools/c2_bench.py > $OUT/c2.json 2> $OUT/c2.err; echo "c2 rc=$?"; head -c 3000 $OUT/c2.json; tail -3 $OUT/c2.err
timeout 900 python bench.py --steps 5 --warmup 2 --no-inference --no-frontend --no-baseline-c2 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; head -c 1500 $OUT/bench.json; tail -3 $OUT/bench.err
python - <<'PY'
import json
try:
    j = json.load(open("gpurun_out/r04a/bench.json"))
    print("ms_per_step", j["ms_per_step"], "parity", j.get("parity_check"))
    print("cpu", json.dumps(j.get("cpu_baseline"))[:1500]); print("speedup", j.get("speedup_vs_cpu_baseline"))
except Exception as e: print("no bench json", e)
PY
