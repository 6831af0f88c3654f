#!/bin/bash
# A/B: weight gradients of the FFT blocks deferred to a side stream (MTTS_DEFER_WGRAD), single-task rank / 2-task rank / 8 tasks
OUT=gpurun_out/r02t; mkdir -p $OUT; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build_device()" > $OUT/build.log 2>&1
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_edge_cases.py -m gpu -q -x > $OUT/pytest.log 2>&1; grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl" $OUT/pytest.log | tail -2
Q="--steps 10 --warmup 3 --no-cpu-baseline --no-inference --no-frontend --no-baseline-c2 --no-bf16x3-leg --no-roofline"
for rep in 1 2; do
for f in 0 1; do
MTTS_DEFER_WGRAD=$f timeout 200 python bench.py $Q --emulate-world 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('defer $f w8 ms', d['ms_per_step'], 'so', d.get('second_order',{}).get('ms_per_step'))"
MTTS_DEFER_WGRAD=$f timeout 200 python bench.py $Q --emulate-world 4 --no-second-order 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('defer $f w4 ms', d['ms_per_step'])"
done
done
MTTS_DEFER_WGRAD=1 timeout 200 python bench.py $Q --no-second-order 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('defer 1 w1 ms', d['ms_per_step'])"
