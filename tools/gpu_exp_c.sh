#!/bin/bash
OUT=gpurun_out/r02c; mkdir -p $OUT; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build_device()" > $OUT/build.log 2>&1
Q="--steps 3 --warmup 1 --no-cpu-baseline --no-inference --no-baseline-c2 --no-bf16x3-leg --no-second-order"
MTTS_GEMM_DUMP=$OUT/sites8.csv timeout 200 python bench.py $Q > $OUT/b8.json 2>/dev/null; python tools/gemm_sites.py $OUT/sites8.csv > $OUT/sites8.md
MTTS_GEMM_DUMP=$OUT/sites1.csv timeout 200 python bench.py $Q --emulate-world 8 > $OUT/b1.json 2>/dev/null; python tools/gemm_sites.py $OUT/sites1.csv > $OUT/sites1.md
timeout 300 python -m pytest tests/test_gpu_c5_training.py -m gpu -q -x -k "feature_tree" > $OUT/pytest.log 2>&1; tail -2 $OUT/pytest.log
head -40 $OUT/sites8.md
