#!/usr/bin/env python3
"""K-loop variants (MTTS_KLOOP, csrc/gemm.h gemm_f32_kloop KL) of the 64x64 BK = 32 kernels: (1) `hash`: the outputs of NT / NN / TN
products (partial last K-slice, ragged M / N) and of a k = 9 input-gradient conv (NN with taps) as a digest — equal digests across
variants = bit-identical results; (2) `time`: TFLOP/s of the three forms on model shapes.  One variant per process (the switch is read
once): tools/kloop_ab.sh loops over them."""
import ctypes as C
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from meta_tts_amd import _lib  # noqa: E402

lib = _lib.load()
P = lambda t: C.c_void_p(t.data_ptr())
g = torch.Generator(device="cpu").manual_seed(5)
R = lambda *s: torch.randn(*s, generator=g).cuda()


def gemm(form, M, N, K, tile=3064):
    if form == 0:
        A, B, lda, ldb = R(M, K), R(N, K), K, K
    elif form == 1:
        A, B, lda, ldb = R(M, K), R(K, N), K, N
    else:
        A, B, lda, ldb = R(K, M), R(K, N), M, N
    Cm = torch.zeros(M, N, device="cuda")
    assert lib.mtts_gemm_f32(form, M, N, K, P(A), lda, P(B), ldb, P(Cm), N, None, 1.0, 0, tile, None) == 0
    return A, B, lda, ldb, Cm


if sys.argv[1] == "hash":
    h = hashlib.sha256()
    for form in (0, 1, 2):
        for (M, N, K) in ((300, 200, 2320), (64, 64, 32), (129, 70, 96), (1000, 256, 9216)):
            h.update(gemm(form, M, N, K)[4].cpu().numpy().tobytes())
    # k = 9 conv forward (mode 0: NT over overlapping rows), input gradient (mode 1: NN, taps walked backwards) and weight gradient (mode 2: TN)
    L, Cin, Cout, k = 500, 256, 1024, 9
    x, w, dy = R(L + 8, Cin), R(Cout, k * Cin), R(L + 8, Cout)
    x[:4] = 0; x[-4:] = 0; dy[:4] = 0; dy[-4:] = 0
    for mode, a, b, shape in ((0, x[4:], w, (L, Cout)), (1, dy[4:], w, (L, Cin)), (2, dy[4:], x[4:], (Cout, k * Cin))):
        out = torch.zeros(*shape, device="cuda")
        rc = lib.mtts_conv1d_f32(mode, L, Cin, Cout, k, P(a), P(b), P(out), None, 3064, None)
        assert rc == 0, rc
        h.update(out.cpu().numpy().tobytes())
    print("KLOOP", os.environ.get("MTTS_KLOOP", "default"), "digest", h.hexdigest()[:16])
else:
    for name, form, M, N, K in (("NT k=9 fwd 8 tasks", 0, 17047, 1024, 2304), ("NT postnet fwd", 0, 22132, 512, 2560), ("NN k=9 dgrad-shaped", 1, 17047, 256, 9216),
                                ("TN k=9 wgrad-shaped", 2, 2304, 1024, 17047), ("NT 4096^3", 0, 4096, 4096, 4096), ("NT short K (qkv)", 0, 17047, 768, 256)):
        A, B, lda, ldb, Cm = gemm(form, M, N, K)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        e0.record()
        for _ in range(reps):
            lib.mtts_gemm_f32(form, M, N, K, P(A), lda, P(B), ldb, P(Cm), N, None, 1.0, 0, 3064, None)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        tf = 2.0 * M * N * K / (ms * 1e-3) / 1e12
        print(f"KLOOP {os.environ.get('MTTS_KLOOP', 'default')}  {name:24s} {1e3 * ms:8.1f} us  {tf:6.1f} TFLOP/s  {tf / 157.3:.3f}")
