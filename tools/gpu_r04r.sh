#!/bin/bash
# round 4, call r: conv tap geometry hoisted out of the GEMM K-loops (no scalar kernel-argument loads per slice): clock-normalised GEMM
# probes vs gpurun_out/r04mid/clock_probe.txt, GEMM kernel parity, 8-task / single-task steps.  -> gpurun_out/r04r/
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r04r; mkdir -p $OUT; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build_device()" > $OUT/build.log 2>&1
rm -f tools/clock_probe; hipcc --offload-arch=gfx950 -O3 tools/clock_probe.hip -Iinclude -Lmeta_tts_amd -lmtts -Wl,-rpath,'$ORIGIN/../meta_tts_amd' -o tools/clock_probe > /dev/null 2>&1
for cfg in "17047 1024 2304 3064" "17047 1024 2304 1064" "4096 4096 4096 3064" "22132 512 2560 3064" "2100 256 2304 4064"; do ./tools/clock_probe $cfg; done 2>&1 | tee $OUT/clock_probe.txt
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m "gpu" -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log; tail -2 $OUT/pytest.log
timeout 1500 python tools/ab.py --world8 --so --steps 6 "BASE" "BASE" > $OUT/ab.txt 2>&1; cat $OUT/ab.txt
