#!/bin/bash
# round 3, run M: sustained MFMA rate / shader clock of the box, and the clock while bench.py runs
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r03m; mkdir -p $OUT
./tools/mfma_peak 200000 2>&1 | tee $OUT/mfma_peak.txt
rocm-smi --showclocks --showpower 2>&1 | grep -v "^$" | head -30 > $OUT/smi_idle.txt
(python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-inference --no-frontend --no-baseline-c2 --no-second-order > $OUT/bench.json 2> $OUT/bench.err) &
BP=$!
sleep 45
for i in 1 2 3 4 5 6 7 8; do rocm-smi --showclocks --showpower 2>&1 | grep -E "sclk|Power|mclk|fclk" >> $OUT/smi_busy.txt; echo "--" >> $OUT/smi_busy.txt; sleep 0.7; done
wait $BP
cat $OUT/smi_busy.txt | head -60
python -c "
import json; d=json.load(open('$OUT/bench.json')); print(d['ms_per_step'], d['roofline']['frac'], d['roofline']['all_gemm']['frac'])"
