#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r03h; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu 2>&1 | tail -6 > $OUT/kernels.txt; cat $OUT/kernels.txt
BENCH_TILES=3064,4064,6064 BENCH_SHAPES="dec 1 task,1task" timeout 300 python tools/gemm_bench.py 2>/dev/null > $OUT/mb.log; cat $OUT/mb.log
timeout 1200 python tools/ab.py --only-world8 --so "MTTS_WIDE_MAX_WGS=0" "BASE" "MTTS_WIDE_MAX_WGS=512" "MTTS_WIDE_MAX_WGS=160" > $OUT/ab.log 2>&1; cat $OUT/ab.log
timeout 600 python tools/ab.py "MTTS_WIDE_MAX_WGS=0" "BASE" >> $OUT/ab.log 2>&1; tail -4 $OUT/ab.log
