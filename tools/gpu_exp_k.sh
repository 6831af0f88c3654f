#!/bin/bash
OUT=gpurun_out/r02r; mkdir -p $OUT; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build_device()" > $OUT/build.log 2>&1
timeout 600 python -m pytest tests/test_gpu_model.py tests/test_kernel_entries.py tests/test_edge_cases.py -m gpu -q -x > $OUT/pytest.log 2>&1; grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl" $OUT/pytest.log | tail -2
Q="--steps 10 --warmup 3 --no-cpu-baseline --no-inference --no-frontend --no-baseline-c2 --no-bf16x3-leg --no-second-order --no-roofline"
for rep in 1 2; do
timeout 200 python bench.py $Q 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('step ms', d['ms_per_step'])"
timeout 200 python bench.py $Q --emulate-world 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('w8 ms', d['ms_per_step'])"
done
