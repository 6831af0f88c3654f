#!/bin/bash
# round 4, call d: fused attention v2 (register-prefetched, two workgroups per CU), LayerNorm-backward partial fold, dropout folds, query
# encoder run-ahead: parity on hardware + A/B of each.   -> gpurun_out/r04d/
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r04d; mkdir -p $OUT; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build_device()" > $OUT/build.log 2>&1
timeout 1200 python -m pytest tests/test_kernel_entries.py tests/test_gpu_model.py tests/test_gpu_timed_config.py tests/test_gpu_c5_training.py tests/test_deferred_paths.py tests/test_edge_cases.py -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log; tail -5 $OUT/pytest.log
timeout 1500 python tools/ab.py --world8 --so --steps 6 "BASE" "MTTS_FUSED_ATTN=0" "MTTS_ENC_AHEAD_QUERY=0" "BASE" "MTTS_FUSED_ATTN=0" > $OUT/ab.txt 2>&1; cat $OUT/ab.txt
X="--no-cpu-baseline --no-inference --no-frontend --no-baseline-c2 --no-second-order"
MTTS_GEMM_DUMP=$OUT/w8_dump.csv timeout 300 python bench.py --steps 3 --warmup 1 --emulate-world 8 $X > $OUT/bench_w8.json 2> $OUT/bench_w8.err
python tools/gemm_sites.py $OUT/w8_dump.csv > $OUT/gemm_sites_1task.md 2>&1; head -50 $OUT/gemm_sites_1task.md
C2_ITERS=10 timeout 300 python tools/c2_bench.py > $OUT/c2.json 2> $OUT/c2.err; python -c "
import json; j=json.load(open('$OUT/c2.json')); print('C2 fp32', j['fp32']['ms_per_step'], 'bf16', j['bf16']['ms_per_step'])"
