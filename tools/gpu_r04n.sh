#!/bin/bash
# round 4, call n: operand planes written by their producers (LayerNorm fwd / bwd, BatchNorm apply / bwd, GEMM epilogues) instead of a
# conversion pass; fused attention in the bf16 mode.  Parity (bf16 + fp32 model tests) + C2.  -> gpurun_out/r04n/
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r04n; mkdir -p $OUT; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build_device()" > $OUT/build.log 2>&1
timeout 1200 python -m pytest tests/test_bf16_mode.py tests/test_gpu_model.py tests/test_kernel_entries.py tests/test_gpu_c5_training.py -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log; tail -2 $OUT/pytest.log
for i in 1 2; do
C2_MODES=fp32,bf16,bf16-staged C2_ITERS=20 timeout 300 python tools/c2_bench.py > $OUT/c2_$i.json 2> $OUT/c2_$i.err; python -c "
import json; j=json.load(open('$OUT/c2_$i.json')); print('C2', {m: j[m]['ms_per_step'] for m in ('fp32','bf16','bf16-staged')}, 'gemm ms', {m: j[m]['roofline']['all_gemm_ms'] for m in ('fp32','bf16','bf16-staged')})"
done
