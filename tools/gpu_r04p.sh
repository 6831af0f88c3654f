#!/bin/bash
# round 4, call p: bf16 mode as it ships (BK = 64 plane kernel, asynchronous shadow refresh, three-launch attention): bf16 + model parity,
# C2 x3, then the 8-task bench line in fp32 (regression check of the twin plumbing on the fp32 path).  -> gpurun_out/r04p/
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r04p; mkdir -p $OUT; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build_device()" > $OUT/build.log 2>&1
timeout 1200 python -m pytest tests/test_bf16_mode.py tests/test_gpu_model.py tests/test_gpu_timed_config.py -m "gpu and not slow" -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log; tail -2 $OUT/pytest.log
for i in 1 2 3; do
C2_MODES=bf16 C2_ITERS=20 timeout 300 python tools/c2_bench.py > $OUT/c2_$i.json 2> $OUT/c2_$i.err; python -c "
import json; j=json.load(open('$OUT/c2_$i.json')); print('C2 bf16', j['bf16']['ms_per_step'], 'gemm ms', j['bf16']['roofline']['all_gemm_ms'])"
done
timeout 1500 python tools/ab.py --world8 --steps 6 "BASE" > $OUT/ab.txt 2>&1; cat $OUT/ab.txt
