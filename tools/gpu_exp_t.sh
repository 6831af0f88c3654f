#!/bin/bash
OUT=gpurun_out/r02zy; mkdir -p $OUT; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build_device()" > $OUT/build.log 2>&1
Q="--steps 10 --warmup 3 --no-cpu-baseline --no-inference --no-frontend --no-baseline-c2 --no-bf16x3-leg --no-roofline --no-second-order"
for v in 0 1; do
MTTS_FWD_SINGLE_MULTI=$v timeout 200 python bench.py $Q --emulate-world 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fwd_multi $v w8 ms', d['ms_per_step'])"
MTTS_FWD_SINGLE_MULTI=$v timeout 200 python bench.py $Q --emulate-world 4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fwd_multi $v w4 ms', d['ms_per_step'])"
done
