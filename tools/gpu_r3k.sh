#!/bin/bash
# round 3, run K: dual-source K-loop + per-step activation sets of second-order MAML — kernel tests, second-order parity tests, A/B timing
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r03k; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_kernel_entries.py -q -m gpu -k "dual" 2>&1 | tail -5 > $OUT/dual_tests.txt; cat $OUT/dual_tests.txt
timeout 1500 python -m pytest tests/test_gpu_timed_config.py tests/test_deferred_paths.py tests/test_gpu_c5_training.py -q -m gpu -x 2>&1 | tail -8 > $OUT/so_tests.txt; cat $OUT/so_tests.txt
timeout 1500 python tools/ab.py --world8 --so --steps 4 "BASE" "MTTS_SO_KEEP_ACT=0" "MTTS_DUAL_SRC=0" "MTTS_SO_KEEP_ACT=0 MTTS_DUAL_SRC=0" 2>&1 | tee $OUT/ab.log
