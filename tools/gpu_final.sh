#!/bin/bash
# Final evidence of the round: GPU tests, the default bench line, the single-task emulation, a kernel trace and the PMC passes
# (HBM bytes + MFMA busy per GEMM kernel, first- and second-order) of the same bench command.
TAG=${1:-r02z}
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
R=$PWD
python -c "import __graft_entry__ as g; g.build_device()" > $OUT/build.log 2>&1
timeout 900 python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log; tail -4 $OUT/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
timeout 300 python bench.py --steps 10 --warmup 3 --emulate-world 8 --no-cpu-baseline --no-inference --no-frontend --no-baseline-c2 --no-bf16x3-leg > $OUT/bench_w8.json 2> $OUT/bench_w8.err
Q="--steps 1 --warmup 0 --no-cpu-baseline --no-inference --no-frontend --no-baseline-c2 --no-bf16x3-leg --no-roofline"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof -o trace -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-inference --no-frontend --no-baseline-c2 --no-bf16x3-leg --no-second-order --no-roofline > $R/$OUT/prof_bench.log 2>&1
for ord in 1 2; do
  i=0
  for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES"; do
    i=$((i+1))
    X=""; [ $ord = 2 ] && X="--order 2"; [ $ord = 1 ] && X="--no-second-order"
    timeout 400 rocprofv3 --pmc $grp -d $R/$OUT/pmc_o${ord}_$i -o pmc -- python $R/bench.py $Q $X > $R/$OUT/pmc_o${ord}_$i.log 2>&1
  done
done
cd $R
DB=$(find $OUT/prof -name "*.db" | head -1); [ -n "$DB" ] && python profiles/summarize_rocpd.py $DB > $OUT/kernel_trace.md
python tools/pmc_to_json.py $OUT/pmc_hbm.json $(find $OUT/pmc_o1_* -name "*.db") > $OUT/pmc1.txt 2>&1
python tools/pmc_to_json.py $OUT/pmc_hbm.json:kernels_second_order $(find $OUT/pmc_o2_* -name "*.db") > $OUT/pmc2.txt 2>&1
find $OUT -name "*.db" -delete
head -c 400 $OUT/bench.json; echo; cat $OUT/pmc_hbm.json | head -40
