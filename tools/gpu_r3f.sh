#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r03f; mkdir -p $OUT
for v in "MTTS_SK_WPE=1" "MTTS_SK_WPE=6"; do
  echo "== $v" >> $OUT/mb.log
  env $v BENCH_TILES=1064,5064 BENCH_SHAPES="conv1_fwd,postnet,dec 1 task" timeout 300 python tools/gemm_bench.py 2>/dev/null >> $OUT/mb.log
done
cat $OUT/mb.log
timeout 1200 python tools/ab.py --world8 "MTTS_SK=0" "MTTS_SK_WPE=1" "MTTS_SK_WPE=1 MTTS_SK_SMAX=1" "MTTS_SK_WPE=1 MTTS_SK_MIN_UNITS=160" > $OUT/ab.log 2>&1; cat $OUT/ab.log
