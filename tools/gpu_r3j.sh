#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r03j; mkdir -p $OUT
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -12 > $OUT/gpu_pytest.txt; cat $OUT/gpu_pytest.txt
timeout 900 python bench.py --steps 10 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err; tail -c 3000 $OUT/bench.json; tail -3 $OUT/bench.err
timeout 600 python bench.py --steps 10 --warmup 3 --emulate-world 8 --no-cpu-baseline --no-inference --no-frontend --no-baseline-c2 > $OUT/bench_w8.json 2>> $OUT/bench.err; python -c "
import json; d=json.load(open('$OUT/bench_w8.json')); print('w8', d['ms_per_step'], d['second_order']['ms_per_step'], d['roofline']['all_gemm'])"
