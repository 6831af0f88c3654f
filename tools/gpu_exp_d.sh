#!/bin/bash
# single-task-per-rank regime: split-K knobs of the stand-alone and batched launchers (whole-step A/B under --emulate-world 8)
OUT=gpurun_out/r02f; mkdir -p $OUT; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build_device()" > $OUT/build.log 2>&1
Q="--steps 8 --warmup 2 --no-cpu-baseline --no-inference --no-baseline-c2 --no-bf16x3-leg --no-roofline --emulate-world 8"
for v in "BASE=1" "MTTS_SPLITK_TARGET=512" "MTTS_SPLITK_TARGET=1024" "MTTS_SPLITK_TARGET=2048" "MTTS_SPLITK_TARGET=1024 MTTS_SPLITK_MINCH=8" "MTTS_SPLITK_TARGET=1024 MTTS_SPLITK_MINCH=4" \
         "MTTS_SPLIT_RATIO=2.5" "MTTS_SPLIT_RATIO=1.0" "MTTS_SPLITK_TARGET=1024 MTTS_SPLIT_RATIO=2.5" "MTTS_GLDS_MAX_WGS=2048" "MTTS_BATCH_MIN_K=0"; do
  echo "== $v" >> $OUT/variants.log
  env $v timeout 200 python bench.py $Q 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fo', d['ms_per_step'], 'so', d['second_order']['ms_per_step'])" >> $OUT/variants.log 2>&1
done
cat $OUT/variants.log
