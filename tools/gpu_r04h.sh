#!/bin/bash
# round 4, call h: B-panel-major XCD tile order for launches of < 8 groups: parity + A/B (single-task rank FO/SO, C2 both numerics modes).  -> gpurun_out/r04h/
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r04h; mkdir -p $OUT; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build_device()" > $OUT/build.log 2>&1
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_kernel_entries.py tests/test_gpu_model.py tests/test_gpu_c5_training.py tests/test_deferred_paths.py tests/test_bf16_mode.py -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log; tail -4 $OUT/pytest.log
timeout 1500 python tools/ab.py --only-world8 --so --steps 6 "BASE" "MTTS_PANEL_ORDER=0" "BASE" "MTTS_PANEL_ORDER=0" > $OUT/ab.txt 2>&1; cat $OUT/ab.txt
X="--no-cpu-baseline --no-inference --no-frontend --no-baseline-c2 --no-second-order"
MTTS_GEMM_DUMP=$OUT/w8_dump.csv timeout 300 python bench.py --steps 3 --warmup 1 --emulate-world 8 $X > $OUT/bench_w8.json 2> $OUT/bench_w8.err
python tools/gemm_sites.py $OUT/w8_dump.csv > $OUT/gemm_sites_1task.md 2>&1; head -16 $OUT/gemm_sites_1task.md
for v in "MTTS_PANEL_ORDER=1" "MTTS_PANEL_ORDER=0"; do env $v C2_ITERS=10 timeout 300 python tools/c2_bench.py > $OUT/c2.json 2> $OUT/c2.err; python -c "
import json; j=json.load(open('$OUT/c2.json')); print('C2 [$v] fp32', j['fp32']['ms_per_step'], 'bf16', j['bf16']['ms_per_step'])"; done
