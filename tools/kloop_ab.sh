#!/bin/bash
# K-loop A/B of round 5: the five variants of the 64x64 BK = 32 fp32 K-loop (csrc/gemm.h: KL, MTTS_KLOOP) —
# bit-identity, stand-alone rates beside the shader clock, the 8-task / single-task bench lines, and the PMC counters per arm.
# usage: tools/kloop_ab.sh [tag]  ->  gpurun_out/<tag>/
TAG=${1:-r05c}
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
R=$PWD
[ -x tools/clock_probe ] || hipcc --offload-arch=gfx950 -O3 tools/clock_probe.hip -Iinclude -Lmeta_tts_amd -lmtts -Wl,-rpath,'$ORIGIN/../meta_tts_amd' -o tools/clock_probe > /dev/null 2>&1
ARMS=${ARMS:-"0 1 2 3 4"}
for kl in $ARMS; do MTTS_KLOOP=$kl timeout 120 python tools/kloop_forms.py hash; done 2>&1 | grep -v amdgpu.ids | tee $OUT/digests.txt
for kl in $ARMS; do
  echo "== MTTS_KLOOP=$kl"
  for cfg in "17047 1024 2304 3064" "22132 512 2560 3064"; do MTTS_KLOOP=$kl timeout 60 ./tools/clock_probe $cfg | grep -v idle; done
  MTTS_KLOOP=$kl timeout 120 python tools/kloop_forms.py time 2>&1 | grep KLOOP
done 2>&1 | tee $OUT/standalone.txt
timeout 900 python tools/ab.py --world8 --steps 6 $(for kl in $ARMS; do echo "MTTS_KLOOP=$kl"; done) 2>&1 | tee $OUT/ab.txt
if [ -z "$NO_PMC" ]; then
cd /tmp
for kl in ${PMC_ARMS:-0 1 4}; do
  i=0
  for grp in "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_INSTS_LDS" "SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM" "SQ_INSTS_VALU SQ_WAIT_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS"; do
    i=$((i+1))
    MTTS_KLOOP=$kl timeout 180 rocprofv3 --pmc $grp -d "$R/$OUT/pmc_kl${kl}_$i" -o pmc -- python $R/tools/kloop_forms.py time > "$R/$OUT/pmc_kl${kl}_$i.log" 2>&1
  done
  echo "== MTTS_KLOOP=$kl (tools/kloop_forms.py time: NT / NN / TN model shapes, per-dispatch means)"
  python $R/tools/pmc_summary.py $(find $R/$OUT/pmc_kl${kl}_* -name "*.db") x 2>&1 | grep -E "gemm_f32_kernel|SQ_" | grep -v "^pmc"
done 2>&1 | tee $R/$OUT/pmc.txt
fi
cd $R
find $OUT -name "*.db" -delete; find $OUT -name "*.csv" -size +1M -delete
