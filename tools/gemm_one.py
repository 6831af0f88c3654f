#!/usr/bin/env python3
"""Run ONE GEMM configuration a few times (for rocprofv3 --pmc passes).  usage: gemm_one.py form tile M N K [reps]"""
import ctypes as C, sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from meta_tts_amd import _lib
lib = _lib.load()
form, tile, M, N, K = [int(x) for x in sys.argv[1:6]]
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 3
P = lambda t: C.c_void_p(t.data_ptr())
if form == 0: A = torch.randn(M, K, device="cuda"); B = torch.randn(N, K, device="cuda"); lda, ldb = K, K
elif form == 1: A = torch.randn(M, K, device="cuda"); B = torch.randn(K, N, device="cuda"); lda, ldb = K, N
else: A = torch.randn(K, M, device="cuda"); B = torch.randn(K, N, device="cuda"); lda, ldb = M, N
Cm = torch.empty(M, N, device="cuda")
for _ in range(reps):
    lib.mtts_gemm_f32(form, M, N, K, P(A), lda, P(B), ldb, P(Cm), N, None, 1.0, 0, tile, None)
torch.cuda.synchronize()
