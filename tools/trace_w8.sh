#!/bin/bash
# kernel trace + critical-stream timeline of the single-task rank (bench.py --emulate-world 8): tools/trace_w8.sh OUT [env ...]
cd "${GRAFT_REPO_ROOT:-/root/repo}"; OUT=gpurun_out/${1:-r06tw}; shift; mkdir -p $OUT; export TMPDIR=/tmp; R=$PWD
X="--no-cpu-baseline --no-inference --no-frontend --no-baseline-c2 --no-second-order --no-roofline --emulate-world ${WORLD:-8} --steps 5 --warmup 1"
cd /tmp
timeout 600 env "$@" rocprofv3 --kernel-trace --stats -d $R/$OUT/prof -o trace -- python $R/bench.py $X > $R/$OUT/prof.log 2>&1
cd $R
DB=$(find $OUT/prof -name "*.db" | head -1)
[ -n "$DB" ] && python profiles/summarize_rocpd.py $DB > $OUT/kernel_trace.md && python tools/timeline.py $DB 0.3 > $OUT/timeline.txt 2>&1
find $OUT -name "*.db" -delete
cat $OUT/timeline.txt
