#!/bin/bash
# round 3, run N: task-per-XCD schedule of the 8-task launches — parity, A/B timing, HBM-side traffic
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r03n; mkdir -p $OUT
R=$PWD
timeout 1200 python -m pytest tests/test_xcd_schedule.py tests/test_gpu_timed_config.py -q -m gpu -x -k "schedule or eight_grouped or equal_the_same" 2>&1 | tail -6 > $OUT/tests.txt; cat $OUT/tests.txt
timeout 1200 python tools/ab.py --so --steps 5 "MTTS_XCD_SCHED=1" "MTTS_XCD_SCHED=0" "MTTS_XCD_SCHED=1" "MTTS_XCD_SCHED=0" 2>&1 | tee $OUT/ab.log
X="--no-cpu-baseline --no-inference --no-frontend --no-baseline-c2"
Q="--steps 1 --warmup 0 $X --no-roofline --no-second-order"
cd /tmp
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 400 rocprofv3 --pmc $grp -d $R/$OUT/pmc_$i -o pmc -- python $R/bench.py $Q > $R/$OUT/pmc_$i.log 2>&1
done
cd $R
python tools/pmc_to_json.py $OUT/pmc_hbm_sched.json $(find $OUT/pmc_* -name "*.db") > $OUT/pmc.txt 2>&1
find $OUT -name "*.db" -delete
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03n/pmc_hbm_sched.json'))['kernels']
for k,v in d.items(): print(k, v.get('hbm_bytes_per_launch'), v.get('launches_sampled'))
PY
