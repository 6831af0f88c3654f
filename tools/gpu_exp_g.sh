#!/bin/bash
# A/B of the bias-gradient column sums fused into the weight-gradient GEMM (MTTS_FUSE_COLSUM)
OUT=gpurun_out/r02j; mkdir -p $OUT; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build_device()" > $OUT/build.log 2>&1
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_gpu_c5_training.py -m gpu -q -x > $OUT/pytest.log 2>&1; tail -2 $OUT/pytest.log
Q="--steps 8 --warmup 2 --no-cpu-baseline --no-inference --no-baseline-c2 --no-bf16x3-leg --no-second-order"
for f in 0 1; do
MTTS_FUSE_COLSUM=$f timeout 200 python bench.py $Q 2>/dev/null > $OUT/b8_$f.json; python -c "import json; d=json.load(open('$OUT/b8_$f.json')); print('fuse $f step ms', d['ms_per_step'], 'dom frac', d['roofline']['frac'], 'all', d['roofline']['all_gemm']['frac'])"
MTTS_FUSE_COLSUM=$f timeout 200 python bench.py $Q --emulate-world 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fuse $f w8 ms', d['ms_per_step'])"
done
