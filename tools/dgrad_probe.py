#!/usr/bin/env python3
"""Is the conv1 input-gradient slower in its NN form (B = the [Cout][k][Cin] weight image walked backwards by taps) than the same
contraction as an NT implicit GEMM over a transposed weight copy [Cin][k][Cout] (both operands K-contiguous, like the forward)?
mtts_conv1d_f32 mode 1 (dgrad, Cin = 256, Cout = 1024, k = 9) vs mode 0 (forward with Cin = 1024, Cout = 256: the same M x 256 x 9216
contraction in NT form)."""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from meta_tts_amd import _lib
lib = _lib.load()
P = lambda t: C.c_void_p(t.data_ptr())


def run(mode, L, cin, cout, k, tile, reps=10):
    G = 4
    if mode == 0:
        a = torch.randn(L + 2 * G, cin, device="cuda"); a[:G] = 0; a[-G:] = 0
        b = torch.randn(cout, k, cin, device="cuda")
        out = torch.empty(L, cout, device="cuda")
        args = (mode, L, cin, cout, k, C.c_void_p(a.data_ptr() + 4 * G * cin), P(b), P(out), None, tile, None)
    else:
        a = torch.randn(L + 2 * G, cout, device="cuda"); a[:G] = 0; a[-G:] = 0
        b = torch.randn(cout, k, cin, device="cuda")
        out = torch.empty(L, cin, device="cuda")
        args = (mode, L, cin, cout, k, C.c_void_p(a.data_ptr() + 4 * G * cout), P(b), P(out), None, tile, None)
    assert lib.mtts_conv1d_f32(*args) == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        lib.mtts_conv1d_f32(*args)
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps


for L in (1955, 6900):
    for tile in (0, 3064, 4064):
        nn = run(1, L, 256, 1024, 9, tile)
        nt = run(0, L, 1024, 256, 9, tile)
        print(f"rows {L:5d} tile {tile:4d}: dgrad NN (taps) {nn:7.1f} us   same contraction NT {nt:7.1f} us   ratio {nn / nt:.2f}")
    nn = run(1, L, 512, 512, 5, 4064); nt = run(0, L, 512, 512, 5, 4064)
    print(f"rows {L:5d} postnet k=5 512->512 tile 4064: NN {nn:7.1f} us  NT {nt:7.1f} us  ratio {nn / nt:.2f}")
