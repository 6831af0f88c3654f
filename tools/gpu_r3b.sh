#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r03b; mkdir -p $OUT
for v in "MTTS_SK_WPE=5" "MTTS_SK_WPE=4"; do
  echo "== $v" >> $OUT/mb.log
  env $v BENCH_TILES=1064,3064,5064 BENCH_SHAPES="conv1_fwd,dec 1 task" timeout 300 python tools/gemm_bench.py 2>/dev/null >> $OUT/mb.log
done
cat $OUT/mb.log
(cd /tmp && BENCH_TILES=1064,5064 BENCH_SHAPES="dec 1 task" timeout 300 rocprofv3 --kernel-trace --stats -d $OLDPWD/$OUT/prof_mb -o mb -- python $OLDPWD/tools/gemm_bench.py > $OLDPWD/$OUT/prof_mb.log 2>&1)
python profiles/summarize_rocpd.py $(find $OUT/prof_mb -name "*.db" | head -1) > $OUT/trace_mb.md; cat $OUT/trace_mb.md
(cd /tmp && MTTS_SK_WPE=4 BENCH_TILES=1064,5064 BENCH_SHAPES="dec 1 task" timeout 300 rocprofv3 --kernel-trace --stats -d $OLDPWD/$OUT/prof_mb4 -o mb -- python $OLDPWD/tools/gemm_bench.py > $OLDPWD/$OUT/prof_mb4.log 2>&1)
python profiles/summarize_rocpd.py $(find $OUT/prof_mb4 -name "*.db" | head -1) > $OUT/trace_mb4.md; cat $OUT/trace_mb4.md
