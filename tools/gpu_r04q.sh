#!/bin/bash
# round 4, call q: BK = 64 plane K-loop inside the multi-problem launches (gemm_bf16_multi_planes_kernel): parity + C2 A/B.  -> gpurun_out/r04q/
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r04q; mkdir -p $OUT; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build_device()" > $OUT/build.log 2>&1
timeout 1200 python -m pytest tests/test_bf16_mode.py -m "gpu" -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log; tail -2 $OUT/pytest.log
for v in "X=0" "MTTS_PLANE_BK=32" "X=0" "MTTS_PLANE_BK=32"; do
env $v C2_MODES=bf16 C2_ITERS=20 timeout 300 python tools/c2_bench.py > $OUT/c2.json 2> $OUT/c2.err; python -c "
import json; j=json.load(open('$OUT/c2.json')); print('C2 [$v]', j['bf16']['ms_per_step'], 'gemm ms', j['bf16']['roofline']['all_gemm_ms']); print({k: (v['launches'], v['ms']) for k, v in j['bf16']['roofline']['per_kernel'].items()})"
done
