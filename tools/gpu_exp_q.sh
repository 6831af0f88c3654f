#!/bin/bash
# A/B: encoder run-ahead on a second side stream (MTTS_ENC_AHEAD)
OUT=gpurun_out/r02x; mkdir -p $OUT; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build_device()" > $OUT/build.log 2>&1
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_edge_cases.py tests/test_gpu_c5_training.py -m gpu -q -x > $OUT/pytest.log 2>&1; grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl" $OUT/pytest.log | tail -1
Q="--steps 10 --warmup 3 --no-cpu-baseline --no-inference --no-frontend --no-baseline-c2 --no-bf16x3-leg --no-roofline --no-second-order"
for v in 0 1; do
MTTS_ENC_AHEAD=$v timeout 200 python bench.py $Q --emulate-world 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ahead $v w8 ms', d['ms_per_step'])"
MTTS_ENC_AHEAD=$v timeout 200 python bench.py $Q --emulate-world 4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ahead $v w4 ms', d['ms_per_step'])"
MTTS_ENC_AHEAD=$v timeout 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-frontend --no-baseline-c2 --no-bf16x3-leg --no-roofline --no-second-order 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ahead $v c5', {k:v['ms_per_iter'] for k,v in d['inference_c5'].items() if isinstance(v,dict)})"
done
