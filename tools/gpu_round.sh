#!/bin/bash
# One gpurun call of the round-2 routine: GPU parity tests, the default bench line, the single-task-per-rank emulation and a
# rocprofv3 kernel trace of the bench command.  Everything lands under gpurun_out/$TAG.
TAG=${1:-r02a}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build_device()" > $OUT/build.log 2>&1
timeout 900 python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -5 $OUT/pytest.log
timeout 600 python bench.py --steps 10 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
timeout 300 python bench.py --steps 10 --warmup 3 --emulate-world 8 --no-cpu-baseline --no-inference --no-baseline-c2 --no-bf16x3-leg > $OUT/bench_w8.json 2> $OUT/bench_w8.err
if [ "${2:-}" = "prof" ]; then
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OLDPWD/$OUT/prof -o trace -- python $OLDPWD/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-inference --no-baseline-c2 --no-bf16x3-leg --no-second-order > $OLDPWD/$OUT/prof_bench.log 2>&1 )
  DB=$(find $OUT/prof -name "*.db" | head -1)
  [ -n "$DB" ] && python profiles/summarize_rocpd.py $DB > $OUT/kernel_trace.md 2>>$OUT/prof_bench.log
  find $OUT/prof -name "*.db" -size +20M -delete
fi
head -c 600 $OUT/bench.json; echo
head -c 400 $OUT/bench_w8.json; echo
