#!/bin/bash
# Single-task rank (--emulate-world 8): where the step's time is.  (a) default placement, (b) everything on ONE stream (no deferred weight
# gradients, no run-ahead, no side-stream predictors, monolithic SGD step): per-launch GEMM records of both + kernel traces.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; OUT=gpurun_out/${1:-r06p1}; mkdir -p $OUT; export TMPDIR=/tmp; R=$PWD
X="--no-cpu-baseline --no-inference --no-frontend --no-baseline-c2 --no-second-order --emulate-world 8"
SER="MTTS_DEFER_WGRAD=0 MTTS_ENC_AHEAD=0 MTTS_ENC_AHEAD_QUERY=0 MTTS_PRED_SIDE=0 MTTS_PRED_EARLY=0 MTTS_UPD_OVERLAP=0 MTTS_SIDE_PRED_ALL=0 MTTS_ENC_AHEAD_ALL=0"
MTTS_BENCH_KEEP_SITES=$R/$OUT/sites_default.csv python bench.py --steps 6 --warmup 2 $X > $OUT/bench_default.json 2> $OUT/bench_default.err
env $SER MTTS_BENCH_KEEP_SITES=$R/$OUT/sites_serial.csv python bench.py --steps 6 --warmup 2 $X > $OUT/bench_serial.json 2> $OUT/bench_serial.err
python tools/gemm_sites.py $OUT/sites_default.csv > $OUT/sites_default.md
python tools/gemm_sites.py $OUT/sites_serial.csv > $OUT/sites_serial.md
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof_serial -o trace -- env $SER python $R/bench.py --steps 5 --warmup 1 $X --no-roofline > $R/$OUT/prof_serial.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof_default -o trace -- python $R/bench.py --steps 5 --warmup 1 $X --no-roofline > $R/$OUT/prof_default.log 2>&1
cd $R
for t in serial default; do
  DB=$(find $OUT/prof_$t -name "*.db" | head -1)
  [ -n "$DB" ] && python profiles/summarize_rocpd.py $DB > $OUT/kernel_trace_$t.md && python tools/timeline.py $DB 0.3 > $OUT/timeline_$t.txt 2>&1
done
find $OUT -name "*.db" -delete; find $OUT -name "*.csv" -size +3M -delete
python -c "
import json
for t in ('default','serial'):
    d=json.load(open('$OUT/bench_%s.json'%t)); print(t, d['ms_per_step'], d['roofline']['all_gemm']['ms_per_meta_step'], d['roofline']['all_gemm']['frac'])
"
head -30 $OUT/sites_serial.md
