#!/bin/bash
# Does the 128x128 tile beat 64x64 once wave quantisation is taken out?  Shapes whose 128-tile count is an exact multiple of 512
# (2 workgroups per CU x 256 CUs) against the model's own ragged shapes; plus the K-loop after the division removal.
OUT=gpurun_out/r02h; mkdir -p $OUT; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build_device()" > $OUT/build.log 2>&1
BENCH_FORMS=0 BENCH_TILES=1064,3064,4064,1128,3128 BENCH_CUSTOM="q2r,16384,1024,2304;q1r,8192,1024,2304;q3r,24576,1024,2304;q2r_n256,32768,256,9216;q1r_n256,16384,256,9216;ragged,17047,1024,2304;q4r_n512,16384,512,2560" timeout 400 python tools/gemm_bench.py > $OUT/gemm_quant.log 2>&1
grep -v amdgpu $OUT/gemm_quant.log
Q="--steps 6 --warmup 2 --no-cpu-baseline --no-inference --no-baseline-c2 --no-bf16x3-leg --no-second-order --no-roofline"
timeout 200 python bench.py $Q 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('step ms', d['ms_per_step'])"
timeout 200 python bench.py $Q --emulate-world 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('w8 ms', d['ms_per_step'])"
