#!/bin/bash
# round 4, call k: encoder run-ahead for launches of any task count (A/B on the 8-task step, first / second order), new fold kernel +
# batched running statistics (parity: kernel entries, model, timed config), by_site_class in the bench line.  -> gpurun_out/r04k/
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r04k; mkdir -p $OUT; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build_device()" > $OUT/build.log 2>&1
timeout 1200 python -m pytest tests/test_kernel_entries.py tests/test_gpu_model.py tests/test_gpu_c5_training.py tests/test_deferred_paths.py tests/test_gpu_timed_config.py -m "gpu and not slow" -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log; tail -4 $OUT/pytest.log
timeout 1500 python tools/ab.py --world8 --so --steps 6 "BASE" "MTTS_ENC_AHEAD_ALL=0" "BASE" "MTTS_ENC_AHEAD_ALL=0" > $OUT/ab.txt 2>&1; cat $OUT/ab.txt
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-inference --no-frontend --no-baseline-c2 --no-second-order > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python - <<PY
import json
try:
    d = json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
    print("ms", d["ms_per_step"]); print("roof", d["roofline"]["frac"], json.dumps(d["roofline"].get("by_site_class"), indent=1))
except Exception as e:
    print("no bench line", e); print(open("$OUT/bench.err").read()[-1500:])
PY
