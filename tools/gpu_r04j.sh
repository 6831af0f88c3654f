#!/bin/bash
# round 4, call j: side-stream predictors for launches of any task count (A/B on the 8-task step), bucket-table / speaker gradients on the
# side stream (single-task rank), default bench line with by_site_class + the concurrent CPU search.  -> gpurun_out/r04j/
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r04j; mkdir -p $OUT; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build_device()" > $OUT/build.log 2>&1
timeout 1200 python -m pytest tests/test_gpu_model.py tests/test_gpu_c5_training.py tests/test_deferred_paths.py tests/test_gpu_timed_config.py -m "gpu and not slow" -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log; tail -4 $OUT/pytest.log
timeout 1500 python tools/ab.py --world8 --so --steps 6 "BASE" "MTTS_SIDE_PRED_ALL=0" "BASE" "MTTS_SIDE_PRED_ALL=0" > $OUT/ab.txt 2>&1; cat $OUT/ab.txt
( time timeout 900 python bench.py --steps 10 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err ) 2> $OUT/bench_time.txt; echo "bench rc=$?"; tail -3 $OUT/bench_time.txt
python - <<PY
import json
try:
    d = json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
    print("ms", d["ms_per_step"], "parity", json.dumps(d.get("parity_check"))[:400])
    print("roof", json.dumps(d["roofline"].get("by_site_class")), d["roofline"]["frac"])
    c = d.get("cpu_baseline"); print("cpu", c["value"], c["leg"], json.dumps(c.get("concurrent"))[:700])
except Exception as e:
    print("no bench line", e); print(open("$OUT/bench.err").read()[-1500:])
PY
