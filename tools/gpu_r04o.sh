#!/bin/bash
# round 4, call o: C2 bf16 A/B — fused attention on/off in the bf16 mode, BK = 64 plane kernel; kernel trace.  -> gpurun_out/r04o/
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r04o; mkdir -p $OUT; export TMPDIR=/tmp
R=$PWD
python -c "import __graft_entry__ as g; g.build_device()" > $OUT/build.log 2>&1
for v in "X=0" "MTTS_FUSED_ATTN=0" "MTTS_PLANE_BK=64" "X=0" "MTTS_FUSED_ATTN=0" "MTTS_PLANE_BK=64"; do
env $v C2_MODES=bf16 C2_ITERS=20 timeout 300 python tools/c2_bench.py > $OUT/c2.json 2> $OUT/c2.err; python -c "
import json; j=json.load(open('$OUT/c2.json')); print('C2 [$v]', j['bf16']['ms_per_step'], 'gemm ms', j['bf16']['roofline']['all_gemm_ms'])"
done
cd /tmp
C2_MODES=bf16 C2_ITERS=20 timeout 600 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof -o trace -- python $R/tools/c2_bench.py > $R/$OUT/prof.log 2>&1
cd $R
DB=$(find $OUT/prof -name "*.db" | head -1)
[ -n "$DB" ] && python profiles/summarize_rocpd.py $DB > $OUT/kernel_trace_c2_bf16.md && python tools/timeline.py $DB 0.3 > $OUT/timeline_c2_bf16.txt 2>&1
find $OUT -name "*.db" -delete; find $OUT -name "*.csv" -size +2M -delete
head -34 $OUT/timeline_c2_bf16.txt
