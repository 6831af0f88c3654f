#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r03g; mkdir -p $OUT
timeout 1500 python tools/ab.py --only-world8 --so "BASE" "MTTS_SPLITK_TARGET=512" "MTTS_SPLITK_TARGET=1024" "MTTS_SPLITK_TARGET=1024 MTTS_SPLITK_MINCH=8" "MTTS_SPLIT_RATIO=2.5" "MTTS_SPLIT_RATIO=4" "MTTS_SPLITK_TARGET=1024 MTTS_SPLIT_RATIO=3" "MTTS_GLDS_MAX_WGS=2048" "MTTS_MULTI_BK=32" > $OUT/ab_w8.log 2>&1; cat $OUT/ab_w8.log
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > $OUT/gpu_pytest.txt; cat $OUT/gpu_pytest.txt
