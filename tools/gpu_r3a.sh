#!/bin/bash
# round 3, GPU call A: kernel tests of the work-queue kernel, micro-benchmark, whole-step A/B, single-task timeline
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r03a; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu 2>&1 | tail -8 > $OUT/kernels.txt
cat $OUT/kernels.txt
BENCH_SHAPES="conv1_fwd,conv1_dgrad,postnet,qkv,dec 1 task" timeout 300 python tools/gemm_bench.py > $OUT/gemm_bench.log 2>&1
cat $OUT/gemm_bench.log
timeout 1500 python tools/ab.py --world8 "MTTS_SK=0" "BASE" "MTTS_SK_WPE=4" "MTTS_SK_BK=32" "MTTS_SK_SMAX=1" "MTTS_SK_MIN_UNITS=24 MTTS_SK_MIN_TILE=16" "MTTS_SK_MIN_UNITS=160" > $OUT/ab.log 2>&1
cat $OUT/ab.log
B="--no-roofline --no-second-order --no-cpu-baseline --no-inference --no-frontend --no-baseline-c2 --steps 4 --warmup 2"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OLDPWD/$OUT/prof_w8 -o w8 -- python $OLDPWD/bench.py $B --emulate-world 8 > $OLDPWD/$OUT/prof_w8.log 2>&1)
DB=$(find $OUT/prof_w8 -name "*.db" | head -1)
python tools/timeline.py $DB 0.55 > $OUT/timeline_w8.md 2>&1
python profiles/summarize_rocpd.py $DB > $OUT/trace_w8.md 2>&1
cat $OUT/timeline_w8.md
find $OUT -name "*.db" -size +40M -delete
