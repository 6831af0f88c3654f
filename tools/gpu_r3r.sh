#!/bin/bash
# round 3, run R: per-call-site GEMM efficiency of the final build (8-task first / second order)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r03r; mkdir -p $OUT
X="--steps 2 --warmup 1 --no-cpu-baseline --no-inference --no-frontend --no-baseline-c2 --no-second-order"
MTTS_GEMM_DUMP=$OUT/fo.csv timeout 300 python bench.py $X > $OUT/fo.json 2> $OUT/fo.err
MTTS_GEMM_DUMP=$OUT/so.csv timeout 300 python bench.py $X --order 2 > $OUT/so.json 2> $OUT/so.err
python tools/gemm_sites.py $OUT/fo.csv > $OUT/sites_fo.md; python tools/gemm_sites.py $OUT/so.csv > $OUT/sites_so.md
head -30 $OUT/sites_fo.md; rm -f $OUT/*.csv
