#!/bin/bash
# Where the fused attention forward spends its time: the kernel with phases switched off (a -DMTTS_ATTN_DIAG build of the library: bit 0 / 1 / 2 of
# MTTS_ATTN_DIAG_MASK skip Q K^T / softmax / P V — wrong results, timing only), kernel durations from rocprofv3 --kernel-trace --stats.
# usage: tools/attn_phases.sh [tag]
cd "${GRAFT_REPO_ROOT:-/root/repo}"; OUT=gpurun_out/${1:-r06attn}; mkdir -p $OUT; export TMPDIR=/tmp; R=$PWD
[ -f tools/libmtts_diag.so ] || hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DMTTS_ATTN_DIAG -Wno-unused-value -Wno-unused-result meta_tts_amd/csrc/mtts.hip -o tools/libmtts_diag.so
export MTTS_PROBE_LIB=$R/tools/libmtts_diag.so
cd /tmp
for shape in "421 80" "589 10" "128 80"; do
  for mask in 0 1 2 4 3 5 6 7 14 22 30; do
    rm -rf /tmp/attn_prof; MTTS_ATTN_DIAG_MASK=$mask rocprofv3 --kernel-trace --stats -d /tmp/attn_prof -o t -- python $R/tools/attn_probe.py $shape 20 > /tmp/attn_prof.log 2>&1
    DB=$(find /tmp/attn_prof -name "*.db" | head -1)
    python - "$DB" "$shape" $mask <<'PY'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, start, end from kernels").fetchall()
d = [(e - s) / 1e3 for n, s, e in rows if "attn_fwd_kernel" in n]
d = d[1:] if len(d) > 1 else d
print(f"shape (L n_mat) {sys.argv[2]:>8}  skip-mask {sys.argv[3]}  launches {len(d)}  avg {sum(d) / max(len(d), 1):8.1f} us  min {min(d) if d else 0:8.1f} us")
PY
  done
done | tee $R/$OUT/attn_phases.txt
