#!/bin/bash
OUT=gpurun_out/r02s; mkdir -p $OUT; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build_device()" > $OUT/build.log 2>&1
Q="--steps 10 --warmup 3 --no-cpu-baseline --no-inference --no-frontend --no-baseline-c2 --no-bf16x3-leg --no-second-order"
for rep in 1 2; do
for bk in 0 32; do
MTTS_MULTI_BK=$bk timeout 200 python bench.py $Q 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('multi_bk $bk step ms', d['ms_per_step'], 'dom', d['roofline']['frac'], 'all', d['roofline']['all_gemm']['frac'])"
done
done
