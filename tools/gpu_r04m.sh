#!/bin/bash
# round 4, call m: bf16 mode split-K fill rule A/B on C2, kernel trace + stream timeline of the C2 bf16 step.  -> gpurun_out/r04m/
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r04m; mkdir -p $OUT; export TMPDIR=/tmp
R=$PWD
python -c "import __graft_entry__ as g; g.build_device()" > $OUT/build.log 2>&1
timeout 900 python -m pytest tests/test_bf16_mode.py -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log; tail -2 $OUT/pytest.log
for v in "MTTS_BF16_SPLIT_TARGET=1536" "MTTS_BF16_SPLIT_TARGET=0" "MTTS_BF16_SPLIT_TARGET=1024" "MTTS_BF16_SPLIT_TARGET=2560"; do
env $v C2_MODES=bf16 C2_ITERS=20 timeout 300 python tools/c2_bench.py > $OUT/c2.json 2> $OUT/c2.err; python -c "
import json; j=json.load(open('$OUT/c2.json')); print('C2 [$v]', j['bf16']['ms_per_step'], 'gemm ms', j['bf16']['roofline']['all_gemm_ms'])"
done
cd /tmp
C2_MODES=bf16 C2_ITERS=20 timeout 600 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof -o trace -- python $R/tools/c2_bench.py > $R/$OUT/prof.log 2>&1
cd $R
DB=$(find $OUT/prof -name "*.db" | head -1)
[ -n "$DB" ] && python profiles/summarize_rocpd.py $DB > $OUT/kernel_trace_c2_bf16.md && python tools/timeline.py $DB 0.3 > $OUT/timeline_c2_bf16.txt 2>&1
find $OUT -name "*.db" -delete; find $OUT -name "*.csv" -size +2M -delete
head -40 $OUT/timeline_c2_bf16.txt
