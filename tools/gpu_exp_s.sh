#!/bin/bash
OUT=gpurun_out/r02zz; mkdir -p $OUT; export TMPDIR=/tmp
R=$PWD
python -c "import __graft_entry__ as g; g.build_device()" > $OUT/build.log 2>&1
Q="--steps 3 --warmup 1 --no-cpu-baseline --no-inference --no-frontend --no-baseline-c2 --no-bf16x3-leg --no-roofline --no-second-order"
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $R/$OUT/fo1 -o t -- python $R/bench.py $Q --emulate-world 8 > $R/$OUT/fo1.log 2>&1
cd $R
DB=$(find $OUT/fo1 -name "*.db" | head -1); [ -n "$DB" ] && python profiles/summarize_rocpd.py $DB > $OUT/fo1.md
find $OUT -name "*.db" -delete
grep -o '"ms_per_step": [0-9.]*' $OUT/fo1.log | head -1; head -12 $OUT/fo1.md | cut -c1-120; tail -1 $OUT/fo1.md
