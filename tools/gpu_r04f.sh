#!/bin/bash
# round 4, call f: after the knob pruning + attention with pinned prefetch: the whole GPU suite, A/B of the fused attention.  -> gpurun_out/r04f/
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r04f; mkdir -p $OUT; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build_device()" > $OUT/build.log 2>&1
timeout 1500 python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log; tail -5 $OUT/pytest.log
timeout 1500 python tools/ab.py --world8 --so --steps 6 "BASE" "MTTS_FUSED_ATTN=0" "BASE" "MTTS_FUSED_ATTN=0" > $OUT/ab.txt 2>&1; cat $OUT/ab.txt
X="--no-cpu-baseline --no-inference --no-frontend --no-baseline-c2 --no-second-order"
MTTS_GEMM_DUMP=$OUT/w8_dump.csv timeout 300 python bench.py --steps 3 --warmup 1 --emulate-world 8 $X > $OUT/bench_w8.json 2> $OUT/bench_w8.err
python tools/gemm_sites.py $OUT/w8_dump.csv > $OUT/gemm_sites_1task.md 2>&1; grep "attn" $OUT/gemm_sites_1task.md
