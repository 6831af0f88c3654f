#!/bin/bash
# round 4, call c: the fused attention forward — parity tests on hardware, A/B against the three-launch form (8-task step, single-task rank,
# first + second order), per-site GEMM dump of a single-task rank.   -> gpurun_out/r04c/
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r04c; mkdir -p $OUT; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build_device()" > $OUT/build.log 2>&1
timeout 900 python -m pytest tests/test_kernel_entries.py tests/test_gpu_model.py tests/test_gpu_timed_config.py -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log; tail -5 $OUT/pytest.log
timeout 1200 python tools/ab.py --world8 --so --steps 6 "BASE" "MTTS_FUSED_ATTN=0" "BASE" "MTTS_FUSED_ATTN=0" > $OUT/ab_attn.txt 2>&1; cat $OUT/ab_attn.txt
X="--no-cpu-baseline --no-inference --no-frontend --no-baseline-c2 --no-second-order"
MTTS_GEMM_DUMP=$OUT/w8_dump.csv timeout 300 python bench.py --steps 3 --warmup 1 --emulate-world 8 $X > $OUT/bench_w8.json 2> $OUT/bench_w8.err
python tools/gemm_sites.py $OUT/w8_dump.csv > $OUT/gemm_sites_1task.md 2>&1; head -60 $OUT/gemm_sites_1task.md
