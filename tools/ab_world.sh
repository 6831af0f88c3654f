#!/bin/bash
# A/B over the ranks of the scaling curve: tools/ab_world.sh OUT "ENV1" "ENV2" ... — for every variant the step of a rank holding 8 / 4 / 2 / 1 tasks
# (bench.py --emulate-world 1 / 2 / 4 / 8), first order (and second order with SO=1).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; OUT=gpurun_out/$1; shift; mkdir -p $OUT
X="--no-cpu-baseline --no-inference --no-frontend --no-baseline-c2 --no-roofline --steps 6 --warmup 2"
[ "${SO:-0}" = "1" ] || X="$X --no-second-order"
for V in "$@"; do
  echo "== $V"
  for W in ${WORLDS:-1 2 4 8}; do
    E=""; [ "$V" = "BASE" ] || E="$V"
    env $E python bench.py $X $EXTRA --emulate-world $W > $OUT/tmp.json 2> $OUT/tmp.err
    python - $OUT/tmp.json $W <<'P'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    so = d.get("second_order") or {}
    print(f"  world {sys.argv[2]} ({8 // int(sys.argv[2])} tasks/rank): {d['ms_per_step']:.2f} ms (host enqueue {d.get('host_enqueue_ms_from_idle')} ms)" + (f"  so {so['ms_per_step']:.2f} ms" if so else ""))
except Exception as ex:
    print("  world", sys.argv[2], "ERROR", ex)
P
  done
done
