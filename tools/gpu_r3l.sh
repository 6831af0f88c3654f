#!/bin/bash
# round 3, run L: kept primal gradients in the second-order reverse sweep — second-order parity tests + A/B timing
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r03l; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_timed_config.py tests/test_deferred_paths.py tests/test_gpu_c5_training.py tests/test_gpu_model.py -q -m gpu -x -k "second_order or so or hessian or imaml or side_stream" 2>&1 | tail -8 > $OUT/so_tests.txt; cat $OUT/so_tests.txt
timeout 1500 python tools/ab.py --world8 --so --steps 4 "BASE" "MTTS_SO_KEEP_GRAD=0" 2>&1 | tee $OUT/ab.log
