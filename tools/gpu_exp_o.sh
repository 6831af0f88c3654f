#!/bin/bash
OUT=gpurun_out/r02v; mkdir -p $OUT; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build_device()" > $OUT/build.log 2>&1
Q="--steps 8 --warmup 2 --no-cpu-baseline --no-inference --no-frontend --no-baseline-c2 --no-bf16x3-leg --no-roofline --no-second-order"
for t in 2 4 8; do
MTTS_DEFER_TASKS=$t MTTS_DEFER_MAX_ROWS=40000 timeout 200 python bench.py $Q --emulate-world 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('defer_tasks $t w2 ms', d['ms_per_step'])"
done
MTTS_DEFER_TASKS=8 MTTS_DEFER_MAX_ROWS=40000 timeout 200 python bench.py $Q 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('defer_tasks 8 w1 ms', d['ms_per_step'])"
timeout 200 python bench.py $Q 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('default w1 ms', d['ms_per_step'])"
