#!/bin/bash
# task-per-XCD tile order of the multi-problem launches: time and L2-miss traffic
OUT=gpurun_out/r02o; mkdir -p $OUT; export TMPDIR=/tmp
R=$PWD
python -c "import __graft_entry__ as g; g.build_device()" > $OUT/build.log 2>&1
timeout 300 python -m pytest tests/test_gpu_model.py -m gpu -q -x 2>&1 | tail -1
Q="--steps 8 --warmup 2 --no-cpu-baseline --no-inference --no-frontend --no-baseline-c2 --no-bf16x3-leg --no-second-order"
P="--steps 1 --warmup 0 --no-cpu-baseline --no-inference --no-frontend --no-baseline-c2 --no-bf16x3-leg --no-second-order --no-roofline"
for m in 0 1 2; do
  MTTS_XCD_TASK=$m timeout 200 python bench.py $Q 2>/dev/null > $OUT/b_$m.json; python -c "import json; d=json.load(open('$OUT/b_$m.json')); print('xcd_task $m step ms', d['ms_per_step'], 'dom frac', d['roofline']['frac'])"
  cd /tmp
  MTTS_XCD_TASK=$m timeout 300 rocprofv3 --pmc FETCH_SIZE -d $R/$OUT/pmc_${m}_1 -o pmc -- python $R/bench.py $P > $R/$OUT/pmc_${m}_1.log 2>&1
  MTTS_XCD_TASK=$m timeout 300 rocprofv3 --pmc WRITE_SIZE -d $R/$OUT/pmc_${m}_2 -o pmc -- python $R/bench.py $P > $R/$OUT/pmc_${m}_2.log 2>&1
  cd $R
  python tools/pmc_to_json.py $OUT/pmc_$m.json $(find $OUT/pmc_${m}_* -name "*.db") > $OUT/pmc_$m.txt 2>&1
  python -c "import json; d=json.load(open('$OUT/pmc_$m.json'))['kernels']; print({k: round(v['hbm_bytes_per_launch']/1e6,1) for k,v in d.items()})"
done
find $OUT -name "*.db" -delete
