#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r04g; mkdir -p $OUT; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build_device()" > $OUT/build.log 2>&1
python tools/dgrad_probe.py 2>&1 | grep -v amdgpu.ids | tee $OUT/dgrad_probe.txt
