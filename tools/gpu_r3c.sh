#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r03c; mkdir -p $OUT
BENCH_TILES=1064,3064,5064,5032 BENCH_SHAPES="conv1_fwd,dec 1 task,postnet" timeout 300 python tools/gemm_bench.py 2>/dev/null > $OUT/mb.log; cat $OUT/mb.log
for tile in 1064 5064; do
 for pass in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAIT_INST_ANY" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_SMEM TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE"; do
  n=$(echo $pass | cut -c1-12 | tr ' ' '_')
  (cd /tmp && timeout 200 rocprofv3 --pmc $pass -d $OLDPWD/$OUT/pmc_${tile}_$n -o p -- python $OLDPWD/tools/gemm_one.py 0 $tile 2100 256 1024 5 > $OLDPWD/$OUT/pmc_${tile}_$n.log 2>&1)
  echo "== tile $tile: $pass" >> $OUT/pmc.txt
  python tools/pmc_summary.py $(find $OUT/pmc_${tile}_$n -name "*.db" | head -1) >> $OUT/pmc.txt 2>&1
 done
done
cat $OUT/pmc.txt
timeout 900 python tools/ab.py --world8 "MTTS_SK=0" "BASE" "MTTS_SK_SMAX=1" "MTTS_SK_MIN_UNITS=160" > $OUT/ab.log 2>&1; cat $OUT/ab.log
