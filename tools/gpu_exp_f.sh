#!/bin/bash
OUT=gpurun_out/r02i; mkdir -p $OUT; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build_device()" > $OUT/build.log 2>&1
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_vocoder.py -m gpu -q -x > $OUT/pytest.log 2>&1; tail -2 $OUT/pytest.log
BENCH_TILES=1064,3064,4064,1128 BENCH_SHAPES="conv1_fwd,conv1_dgrad,qkv,dec 1 task,postnet" timeout 300 python tools/gemm_bench.py 2>/dev/null > $OUT/gemm_bench.log; cat $OUT/gemm_bench.log
Q="--steps 8 --warmup 2 --no-cpu-baseline --no-inference --no-baseline-c2 --no-bf16x3-leg --no-second-order"
timeout 200 python bench.py $Q 2>/dev/null > $OUT/b8.json; python -c "import json; d=json.load(open('$OUT/b8.json')); print('step ms', d['ms_per_step'], 'dom frac', d['roofline']['frac'], 'all', d['roofline']['all_gemm']['frac'])"
timeout 200 python bench.py $Q --emulate-world 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('w8 ms', d['ms_per_step'])"
