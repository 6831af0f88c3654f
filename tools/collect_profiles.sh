#!/bin/bash
# Copy the judged summaries of an evidence run (tools/gpu_final.sh <tag>) from gpurun_out/<tag>/ into profiles/ under per-round names.
# usage: tools/collect_profiles.sh <tag> [round prefix, default r06]
T=${1:-r06fin}; P=${2:-r06}; S=gpurun_out/$T; D=profiles
cp $S/bench.json $D/${P}_bench_final.json
cp $S/bench_w8.json $D/${P}_bench_final_emulate_world8.json
cp $S/pmc_hbm.json $D/${P}_pmc_hbm.json
cp $S/clock_probe.txt $D/${P}_clock_probe.txt
hdr() { echo "<!-- $1 (1x MI355X, tools/gpu_final.sh $T; summarised by profiles/summarize_rocpd.py from the rocprofv3 --kernel-trace --stats database) -->"; }
{ hdr "rocprofv3 --kernel-trace --stats -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-inference --no-frontend --no-baseline-c2 --no-roofline --no-second-order: 8-task first-order meta-step"; cat $S/kernel_trace_8.md; } > $D/${P}_kernel_trace_final.md
{ hdr "... --steps 3 --order 2: 8-task second-order meta-step"; cat $S/kernel_trace_8so.md; } > $D/${P}_kernel_trace_second_order.md
{ hdr "... --steps 5 --emulate-world 8: single-task rank, first order"; cat $S/kernel_trace_1.md; } > $D/${P}_kernel_trace_single_task.md
{ hdr "... --steps 3 --emulate-world 8 --order 2: single-task rank, second order"; cat $S/kernel_trace_1so.md; } > $D/${P}_kernel_trace_single_task_second_order.md
for t in 8 8so 1 1so; do cp $S/timeline_$t.txt $D/${P}_timeline_$t.txt; done
{ echo "# python -m pytest tests -m gpu -q on the MI355X box (tools/gpu_final.sh $T)"; grep -vE "^(HIP version|ROCm version|Hostname|Librccl path|RCCL version|.*NCCL_DEBUG)" $S/pytest.log | tail -n 30; echo; cat $S/smoke.log | tail -n 2; } > $D/${P}_gpu_pytest_final.txt
[ -f $S/kernel_trace_c2_bf16.md ] && { hdr "rocprofv3 --kernel-trace --stats -- python tools/c2_bench.py (C2_MODES=bf16): BASELINE config C2, bf16 numerics mode"; cat $S/kernel_trace_c2_bf16.md; } > $D/${P}_kernel_trace_c2_bf16.md
[ -f $S/emulate_world.txt ] && cp $S/emulate_world.txt $D/${P}_emulate_world.txt
ls -la $D | grep ${P}_
