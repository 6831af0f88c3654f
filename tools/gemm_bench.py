#!/usr/bin/env python3
"""Micro-benchmark of the grouped fp32-MFMA GEMM through the C ABI (single group): intrinsic TFLOP/s per
form / tile on large squares and on the model's own shapes.  GPU only."""
import ctypes as C
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from meta_tts_amd import _lib

lib = _lib.load()
P = lambda t: C.c_void_p(t.data_ptr())


BF16 = os.environ.get("BENCH_BF16", "0") == "1"   # the bf16 operand family (tile codes 64 / 128) instead of the fp32 kernels


def launch(form, tile, M, N, K, A, lda, B, ldb, Cm):
    if BF16:
        return lib.mtts_gemm_bf16(form, M, N, K, P(A), lda, P(B), ldb, None, None, P(Cm), N, None, 1.0, 0, tile, None)
    return lib.mtts_gemm_f32(form, M, N, K, P(A), lda, P(B), ldb, P(Cm), N, None, 1.0, 0, tile, None)


def bench(form, tile, M, N, K, reps=5):
    if form == 0:
        A = torch.randn(M, K, device="cuda"); B = torch.randn(N, K, device="cuda"); lda, ldb = K, K
    elif form == 1:
        A = torch.randn(M, K, device="cuda"); B = torch.randn(K, N, device="cuda"); lda, ldb = K, N
    else:
        A = torch.randn(K, M, device="cuda"); B = torch.randn(K, N, device="cuda"); lda, ldb = M, N
    Cm = torch.empty(M, N, device="cuda")
    assert launch(form, tile, M, N, K, A, lda, B, ldb, Cm) == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        launch(form, tile, M, N, K, A, lda, B, ldb, Cm)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    return 2.0 * M * N * K / (ms * 1e-3) / 1e12, ms


# tile codes (include/mtts.h): 1064 / 3064 register-staged 64x64 BK16 / BK32, 4064 LDS-DMA, 1128 / 3128 128x128
TILES = tuple(int(t) for t in os.environ.get('BENCH_TILES', '64,128' if BF16 else '3064,4064,3128').split(','))
shapes = [("square4096", 4096, 4096, 4096), ("conv1_fwd 8 tasks", 17047, 1024, 2304), ("conv1_dgrad", 17047, 256, 9216), ("conv2_fwd", 17047, 256, 1024),
          ("qkv", 17047, 768, 256), ("out_proj", 17047, 256, 256), ("postnet_mid", 22132, 512, 2560), ("dec 1 task", 2100, 256, 1024),
          ("1task fc", 1950, 256, 256), ("1task qkv", 1950, 768, 256), ("1task conv1", 1950, 1024, 2304), ("1task dgrad", 1950, 256, 9216), ("1task enc", 424, 256, 768)]
if os.environ.get("BENCH_CUSTOM"):   # "name,M,N,K;name,M,N,K": extra shapes (e.g. tile-quantisation experiments)
    shapes = [(a.split(",")[0], int(a.split(",")[1]), int(a.split(",")[2]), int(a.split(",")[3])) for a in os.environ["BENCH_CUSTOM"].split(";")]
elif os.environ.get("BENCH_SHAPES"):
    shapes = [s for s in shapes if any(k in s[0] for k in os.environ["BENCH_SHAPES"].split(","))]
for name, M, N, K in shapes:
    for form in tuple(int(f) for f in os.environ.get('BENCH_FORMS', '0,1,2').split(',')):
        for tile in TILES:
            if form == 2:  # TN: output [M', N'] small, reduction long — use wgrad-like shapes
                m2, n2, k2 = N, K, M // 8 if M > 8192 else M
            else:
                m2, n2, k2 = M, N, K
            tf, ms = bench(form, tile, m2, n2, k2)
            print(f"{name:20s} form={'NT NN TN'.split()[form]} tile={tile:4d} M={m2:6d} N={n2:5d} K={k2:5d}  {ms:8.3f} ms  {tf:6.1f} TF/s", flush=True)
