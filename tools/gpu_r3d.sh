#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r03e; mkdir -p $OUT
BENCH_TILES=1064,3064,5064,5032 BENCH_SHAPES="conv1_fwd,conv1_dgrad,dec 1 task,postnet" timeout 300 python tools/gemm_bench.py 2>/dev/null > $OUT/mb.log; cat $OUT/mb.log
timeout 1200 python tools/ab.py --world8 "MTTS_SK=0" "BASE" "MTTS_SK_SMAX=1" "MTTS_SK_BK=32" "MTTS_SK_WPE=5" "MTTS_SK_MIN_UNITS=32 MTTS_SK_MIN_TILE=16" "MTTS_SK_MIN_UNITS=160" "MTTS_SK_TOL=4" "MTTS_MULTI_BK=32 MTTS_SK=0" > $OUT/ab.log 2>&1; cat $OUT/ab.log
