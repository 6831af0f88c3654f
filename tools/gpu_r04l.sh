#!/bin/bash
# round 4, call l: bf16 operand planes (weight shadows + activation twins by a convert pass, NT plane-staged K-loop, dgrad as NT over the
# transposed shadow): parity + C2 A/B (bf16 = planes, bf16-staged = rounding in the staging pass only).  -> gpurun_out/r04l/
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r04l; mkdir -p $OUT; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build_device()" > $OUT/build.log 2>&1
timeout 1200 python -m pytest tests/test_bf16_mode.py tests/test_gpu_model.py -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log; tail -4 $OUT/pytest.log
for i in 1 2; do
C2_MODES=fp32,bf16,bf16-staged C2_ITERS=10 MTTS_GEMM_DUMP=$OUT/c2_dump_$i.csv timeout 300 python tools/c2_bench.py > $OUT/c2_$i.json 2> $OUT/c2_$i.err; python -c "
import json; j=json.load(open('$OUT/c2_$i.json')); print('C2', {m: j[m]['ms_per_step'] for m in ('fp32','bf16','bf16-staged')}, 'gemm ms', {m: j[m]['roofline']['all_gemm_ms'] for m in ('fp32','bf16','bf16-staged')})"
done
python tools/gemm_sites.py $OUT/c2_dump_1.csv > $OUT/c2_sites.md 2>&1; head -60 $OUT/c2_sites.md
