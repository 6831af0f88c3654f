#!/bin/bash
# round 3, run O: the bench lines of record (profiles/r03_pmc_hbm.json in place, so roofline.traffic is filled) + rank-share emulation at N = 2, 4, 8
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r03o; mkdir -p $OUT
timeout 900 python bench.py --steps 10 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
X="--no-cpu-baseline --no-inference --no-frontend --no-baseline-c2"
for w in 8 4 2; do
  timeout 400 python bench.py --steps 10 --warmup 3 --emulate-world $w $X > $OUT/bench_w$w.json 2> $OUT/bench_w$w.err
done
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03o/bench.json')); r=d['roofline']
print(d['value'], d['ms_per_step'], r['frac'], r['traffic'], r['traffic_over_alg_bytes'], r['mfma_pipe_busy_frac_pmc'], d['second_order']['ms_per_step'], d['second_order']['roofline'].get('traffic_over_alg_bytes'))
for w in (8,4,2):
    j=json.load(open(f'gpurun_out/r03o/bench_w{w}.json')); print('w',w,j['ms_per_step'], j['second_order']['ms_per_step'], j['roofline']['all_gemm']['frac'])
PY
