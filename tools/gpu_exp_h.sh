#!/bin/bash
# kernel traces of the second-order step (8 tasks and single-task rank) and the first-order single-task rank
OUT=gpurun_out/r02k; mkdir -p $OUT; export TMPDIR=/tmp
R=$PWD
python -c "import __graft_entry__ as g; g.build_device()" > $OUT/build.log 2>&1
Q="--steps 2 --warmup 1 --no-cpu-baseline --no-inference --no-baseline-c2 --no-bf16x3-leg --no-roofline"
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $R/$OUT/so8 -o t -- python $R/bench.py $Q --order 2 > $R/$OUT/so8.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats -d $R/$OUT/so1 -o t -- python $R/bench.py $Q --order 2 --emulate-world 8 > $R/$OUT/so1.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats -d $R/$OUT/fo1 -o t -- python $R/bench.py $Q --no-second-order --emulate-world 8 > $R/$OUT/fo1.log 2>&1
cd $R
for k in so8 so1 fo1; do DB=$(find $OUT/$k -name "*.db" | head -1); [ -n "$DB" ] && python profiles/summarize_rocpd.py $DB > $OUT/$k.md; tail -1 $OUT/$k.log | head -c 300; echo; done
find $OUT -name "*.db" -delete
