#!/bin/bash
# Final evidence of a round: shader clock beside the GEMM kernels, GPU tests, smoke, the default bench line, the single-task emulation,
# kernel traces (8-task and single-task rank, first + second order) and the PMC passes (HBM bytes + MFMA busy per GEMM kernel, first- and
# second-order) of the same bench command.   usage: tools/gpu_final.sh [tag]   ->  gpurun_out/<tag>/
TAG=${1:-r06z}
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
R=$PWD
python -c "import __graft_entry__ as g; g.build_device()" > $OUT/build.log 2>&1
[ -x tools/mfma_peak ] || hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o tools/mfma_peak > /dev/null 2>&1
[ -x tools/clock_probe ] || hipcc --offload-arch=gfx950 -O3 tools/clock_probe.hip -Iinclude -Lmeta_tts_amd -lmtts -Wl,-rpath,'$ORIGIN/../meta_tts_amd' -o tools/clock_probe > /dev/null 2>&1
for cfg in "17047 1024 2304 3064" "17047 1024 2304 1064" "4096 4096 4096 3064" "4096 4096 4096 3128" "22132 512 2560 3064" "2100 256 2304 3064" "2100 256 2304 4064"; do ./tools/clock_probe $cfg; done 2>&1 | tee $OUT/clock_probe.txt
./tools/mfma_peak 100000 2>&1 | tail -3 >> $OUT/clock_probe.txt
timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log; grep -E "passed|failed|^FAILED|pytest rc" $OUT/pytest.log | tail -6
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
X="--no-cpu-baseline --no-inference --no-frontend --no-baseline-c2"
timeout 400 python bench.py --steps 10 --warmup 3 --emulate-world 8 $X > $OUT/bench_w8.json 2> $OUT/bench_w8.err
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof8 -o trace -- python $R/bench.py --steps 5 --warmup 1 $X --no-roofline --no-second-order > $R/$OUT/prof8.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof8so -o trace -- python $R/bench.py --steps 3 --warmup 1 $X --no-roofline --order 2 --no-second-order > $R/$OUT/prof8so.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof1 -o trace -- python $R/bench.py --steps 5 --warmup 1 --emulate-world 8 $X --no-roofline --no-second-order > $R/$OUT/prof1.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof1so -o trace -- python $R/bench.py --steps 3 --warmup 1 --emulate-world 8 $X --no-roofline --order 2 --no-second-order > $R/$OUT/prof1so.log 2>&1
Q="--steps 1 --warmup 0 $X --no-roofline"
for ord in 1 2; do
  i=0
  for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES"; do
    i=$((i+1))
    Y="--no-second-order"; [ $ord = 2 ] && Y="--order 2 --no-second-order"
    timeout 400 rocprofv3 --pmc $grp -d $R/$OUT/pmc_o${ord}_$i -o pmc -- python $R/bench.py $Q $Y > $R/$OUT/pmc_o${ord}_$i.log 2>&1
  done
done
# BASELINE config C2 in the bf16 numerics mode: kernel trace + the same three counter groups (no PMC evidence for any bf16 kernel)
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$OUT/profc2 -o trace -- env C2_MODES=bf16 C2_ITERS=20 python $R/tools/c2_bench.py > $R/$OUT/profc2.log 2>&1
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES"; do
  i=$((i+1)); timeout 300 rocprofv3 --pmc $grp -d $R/$OUT/pmc_c2_$i -o pmc -- env C2_MODES=bf16 C2_ITERS=20 python $R/tools/c2_bench.py > $R/$OUT/pmc_c2_$i.log 2>&1
done
cd $R
# the per-rank step of an N-rank job, N = 2 / 4 / 8 (first and second order)
for w in 2 4 8; do
  timeout 400 python bench.py --steps 6 --warmup 2 --emulate-world $w $X --no-roofline > $OUT/bench_w$w.tmp 2>/dev/null
  python -c "
import json; d=json.load(open('$OUT/bench_w$w.tmp')); print('emulate-world $w tasks/rank', d['config']['tasks_per_gpu'], 'FO ms', d['ms_per_step'], 'SO ms', (d.get('second_order') or {}).get('ms_per_step'))" >> $OUT/emulate_world.txt
done
DB=$(find $OUT/profc2 -name "*.db" | head -1)
[ -n "$DB" ] && python profiles/summarize_rocpd.py $DB > $OUT/kernel_trace_c2_bf16.md
for t in 8 8so 1 1so; do
  DB=$(find $OUT/prof$t -name "*.db" | head -1)
  [ -n "$DB" ] && python profiles/summarize_rocpd.py $DB > $OUT/kernel_trace_$t.md && python tools/timeline.py $DB 0.3 > $OUT/timeline_$t.txt 2>&1
done
python tools/pmc_to_json.py $OUT/pmc_hbm.json $(find $OUT/pmc_o1_* -name "*.db") > $OUT/pmc1.txt 2>&1
python tools/pmc_to_json.py $OUT/pmc_hbm.json:kernels_second_order $(find $OUT/pmc_o2_* -name "*.db") > $OUT/pmc2.txt 2>&1
python tools/pmc_to_json.py $OUT/pmc_hbm.json:kernels_c2_bf16 $(find $OUT/pmc_c2_* -name "*.db") > $OUT/pmc3.txt 2>&1
find $OUT -name "*.db" -delete
find $OUT -name "*.csv" -size +2M -delete
head -c 600 $OUT/bench.json; echo; tail -3 $OUT/prof8.log; head -30 $OUT/pmc_hbm.json
