#!/bin/bash
# BASELINE config C2 (multi-task baseline, batch 16) in the bf16 numerics mode: per-launch GEMM records, kernel trace, timeline, PMC passes (HBM bytes, MFMA busy).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; OUT=gpurun_out/${1:-r06c2}; mkdir -p $OUT; export TMPDIR=/tmp; R=$PWD
python tools/c2_bench.py > $OUT/c2.txt 2>&1; cat $OUT/c2.txt
MTTS_GEMM_DUMP=$R/$OUT/c2_sites.csv C2_MODES=bf16 python tools/c2_bench.py > /dev/null 2>&1; python tools/gemm_sites.py $OUT/c2_sites.csv > $OUT/c2_sites.md 2>/dev/null
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof -o trace -- env C2_MODES=bf16 C2_ITERS=20 python $R/tools/c2_bench.py > $R/$OUT/prof.log 2>&1
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES"; do
  i=$((i+1)); timeout 300 rocprofv3 --pmc $grp -d $R/$OUT/pmc_$i -o pmc -- env C2_MODES=bf16 C2_ITERS=20 python $R/tools/c2_bench.py > $R/$OUT/pmc_$i.log 2>&1
done
cd $R
DB=$(find $OUT/prof -name "*.db" | head -1)
[ -n "$DB" ] && python profiles/summarize_rocpd.py $DB > $OUT/kernel_trace_c2_bf16.md && python tools/timeline.py $DB 0.3 > $OUT/timeline_c2_bf16.txt 2>&1
python tools/pmc_to_json.py $OUT/pmc_c2.json:kernels_c2_bf16 $(find $OUT/pmc_* -name "*.db") > $OUT/pmc.txt 2>&1
find $OUT -name "*.db" -delete; find $OUT -name "*.csv" -size +3M -delete
head -40 $OUT/kernel_trace_c2_bf16.md; head -12 $OUT/timeline_c2_bf16.txt; head -50 $OUT/pmc_c2.json
