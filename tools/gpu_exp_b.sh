#!/bin/bash
# GPU call B: kernel-entry parity on hardware, GEMM variant micro-benchmarks, SQ stall breakdown (PMC) of the 64x64 fp32 kernel,
# and whole-step A/B runs of existing launcher switches.
OUT=gpurun_out/r02b; mkdir -p $OUT; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build_device()" > $OUT/build.log 2>&1
timeout 300 python -m pytest tests/test_kernel_entries.py -m gpu -q -x > $OUT/pytest_kernels.log 2>&1; tail -3 $OUT/pytest_kernels.log
BENCH_TILES=1064,3064,4064,1128,3128 BENCH_SHAPES="conv1_fwd,conv1_dgrad,qkv,dec 1 task" timeout 300 python tools/gemm_bench.py > $OUT/gemm_bench.log 2>&1
Q="--steps 6 --warmup 2 --no-cpu-baseline --no-inference --no-baseline-c2 --no-bf16x3-leg --no-second-order --no-roofline"
for v in "BASE=1" "MTTS_MULTI_BK=32" "MTTS_GLDS_MAX_WGS=1000000" "MTTS_GLDS=0" "MTTS_TILE128_EFF=1.1" "MTTS_XCD_GROUP=0"; do
  echo "== $v" >> $OUT/variants.log
  env $v timeout 200 python bench.py $Q 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])" >> $OUT/variants.log 2>&1
  env $v timeout 200 python bench.py $Q --emulate-world 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('w8', d['ms_per_step'])" >> $OUT/variants.log 2>&1
done
cat $OUT/variants.log
cd /tmp
R=$OLDPWD
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAVES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_INSTS_LDS SQ_INST_CYCLES_VMEM" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $grp -d $R/$OUT/pmc$i -o pmc -- python $R/tools/gemm_one.py 0 1064 17047 1024 2304 6 > $R/$OUT/pmc$i.log 2>&1
  DB=$(find $R/$OUT/pmc$i -name "*.db" | head -1)
  [ -n "$DB" ] && python $R/tools/pmc_summary.py $DB all > $R/$OUT/pmc$i.txt 2>&1
done
cd $R; find $OUT -name "*.db" -size +8M -delete
tail -40 $OUT/gemm_bench.log
