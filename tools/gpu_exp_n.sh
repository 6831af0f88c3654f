#!/bin/bash
OUT=gpurun_out/r02u; mkdir -p $OUT; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build_device()" > $OUT/build.log 2>&1
timeout 600 python -m pytest tests/test_gpu_model.py tests/test_edge_cases.py -m gpu -q -x > $OUT/pytest.log 2>&1; grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl" $OUT/pytest.log | tail -1
Q="--steps 10 --warmup 3 --no-cpu-baseline --no-inference --no-frontend --no-baseline-c2 --no-bf16x3-leg --no-roofline"
for v in 0 1; do
MTTS_DEFER_WGRAD=$v timeout 200 python bench.py $Q --emulate-world 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('defer $v w8 ms', d['ms_per_step'], 'so', d['second_order']['ms_per_step'])"
MTTS_DEFER_WGRAD=$v timeout 200 python bench.py $Q --emulate-world 4 --no-second-order 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('defer $v w4 ms', d['ms_per_step'])"
done
timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-frontend --no-bf16x3-leg --no-roofline --no-second-order 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('w1', d['ms_per_step'], 'c2', d['baseline_c2']['fp32']['ms_per_step'], 'c5', {k:v['ms_per_iter'] for k,v in d['inference_c5'].items() if isinstance(v,dict)})"
