#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r03i; mkdir -p $OUT
BENCH_FORMS=0 BENCH_TILES=3064,4064,6064 BENCH_CUSTOM="m64,64,256,1024;m512,512,256,1024;m2100,2100,256,1024;m8192,8192,256,1024;m64k4096,64,256,4096;m2100k4096,2100,256,4096;m64n64,64,64,1024;m2100n64,2100,64,1024" timeout 300 python tools/gemm_bench.py 2>/dev/null | grep "form=NT" > $OUT/mb.log; cat $OUT/mb.log
