#!/usr/bin/env python3
"""Stand-alone launches of the fused attention forward (csrc/attention.h through mtts_sdpa_fwd) on the decoder's shape of the 8-task step — 80
(sequence, head) groups, L keys each, head width 128 — for `rocprofv3 --pmc` passes and a wall-clock rate.  Usage: attn_probe.py [L] [n_mat] [reps]"""
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from meta_tts_amd import _lib  # noqa: E402

lib = _lib.load(os.environ.get("MTTS_PROBE_LIB"))   # (a diagnostic build: tools/attn_phases.sh)
L = int(sys.argv[1]) if len(sys.argv) > 1 else 421
n_mat = int(sys.argv[2]) if len(sys.argv) > 2 else 80
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
dk = 128
P = lambda t: C.c_void_p(t.data_ptr())
q, k, v = (torch.randn(n_mat, L, dk, device="cuda") for _ in range(3))
ldS = (L + 3) & ~3
Pm = torch.empty(n_mat, L, ldS, device="cuda")
o = torch.empty(n_mat, L, dk, device="cuda")
ws = torch.empty(int(lib.mtts_kernel_ws_bytes(L, n_mat)), dtype=torch.uint8, device="cuda")
assert lib.mtts_sdpa_fwd(n_mat, L, dk, P(q), P(k), P(v), P(Pm), P(o), P(ws), None) == 0
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    lib.mtts_sdpa_fwd(n_mat, L, dk, P(q), P(k), P(v), P(Pm), P(o), P(ws), None)
torch.cuda.synchronize()
us = 1e6 * (time.perf_counter() - t0) / reps
flop = 4.0 * n_mat * L * L * dk
print(f"sdpa_fwd n_mat={n_mat} L={L} dk={dk}: {us:.1f} us per launch, {flop / us * 1e-6:.1f} TFLOP/s = {flop / us * 1e-6 / 157.3:.3f} of 157.3")
