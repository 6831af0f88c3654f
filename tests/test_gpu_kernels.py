"""-m gpu: the MFMA GEMM / implicit-GEMM conv kernels through the C ABI against fp32 torch on the same
seeded inputs (asymmetric operands, ragged sizes so every edge predicate and both tile shapes run)."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as ge
    from meta_tts_amd import _lib
    ge.build_device()
    return _lib.load()


def P(t):
    return C.c_void_p(t.data_ptr())


def _rand(g, *shape):
    return torch.from_numpy(g.standard_normal(shape).astype(np.float32)).cuda()


@pytest.mark.parametrize("tile", [0, 64, 128, 1064, 1128, 2064, 3064, 3128, 4064])
@pytest.mark.parametrize("form", [0, 1, 2])
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (257, 80, 256), (300, 768, 48), (33, 130, 100), (5, 7, 20), (200, 256, 1100), (190, 70, 517)])
def test_gemm_forms(lib, form, tile, M, N, K):
    g = np.random.RandomState(M * 7 + N * 3 + K + form)
    pad4 = lambda x: (x + 3) & ~3
    if form == 0:
        A = _rand(g, M, pad4(K)); B = _rand(g, N, pad4(K)); A[:, K:] = 0; B[:, K:] = 0
        ref = A[:, :K].double() @ B[:, :K].double().T
        lda, ldb = pad4(K), pad4(K)
    elif form == 1:
        A = _rand(g, M, pad4(K)); B = _rand(g, K, pad4(N)); A[:, K:] = 0
        ref = A[:, :K].double() @ B[:, :N].double()
        lda, ldb = pad4(K), pad4(N)
    else:
        A = _rand(g, K, pad4(M)); B = _rand(g, K, pad4(N))
        ref = A[:, :M].double().T @ B[:, :N].double()
        lda, ldb = pad4(M), pad4(N)
    bias = _rand(g, N)
    Cm = torch.full((M, pad4(N)), 7.0, device="cuda")
    rc = lib.mtts_gemm_f32(form, M, N, K, P(A), lda, P(B), ldb, P(Cm), pad4(N), P(bias), 0.5, 0, tile, None)
    assert rc == 0
    torch.cuda.synchronize()
    want = 0.5 * ref + bias.double()[None, :]
    err = (Cm[:, :N].double() - want).abs().max().item()
    assert err < 2e-5 * max(1.0, want.abs().max().item()), err
    assert torch.all(Cm[:, N:] == 7.0)  # nothing written outside [M, N]
    # accumulate + relu flags
    C2 = torch.ones((M, pad4(N)), device="cuda")
    assert lib.mtts_gemm_f32(form, M, N, K, P(A), lda, P(B), ldb, P(C2), pad4(N), None, 1.0, 3, tile, None) == 0
    torch.cuda.synchronize()
    want2 = torch.relu(ref + 1.0)  # accumulate first, then the activation acts on the sum
    assert (C2[:, :N].double() - want2).abs().max().item() < 2e-5 * max(1.0, want2.abs().max().item())


@pytest.mark.parametrize("tile", [0, 64, 1064, 3064, 4064])
@pytest.mark.parametrize("form", [0, 1, 2])
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (257, 80, 256), (33, 130, 100), (200, 256, 1100), (190, 70, 517), (2100, 256, 2304)])
def test_gemm_dual_source(lib, form, tile, M, N, K):
    """C = alpha (A B + A2 B2) + bias in one K-loop chain (the tangent pairs of second-order MAML): every kernel family, the
    split-K path of under-filled launches included, against float64."""
    g = np.random.RandomState(M * 5 + N * 11 + K + form)
    pad4 = lambda x: (x + 3) & ~3

    def operands():
        if form == 0:
            A = _rand(g, M, pad4(K)); B = _rand(g, N, pad4(K)); A[:, K:] = 0; B[:, K:] = 0
            return A, B, A[:, :K].double() @ B[:, :K].double().T, pad4(K), pad4(K)
        if form == 1:
            A = _rand(g, M, pad4(K)); B = _rand(g, K, pad4(N)); A[:, K:] = 0
            return A, B, A[:, :K].double() @ B[:, :N].double(), pad4(K), pad4(N)
        A = _rand(g, K, pad4(M)); B = _rand(g, K, pad4(N))
        return A, B, A[:, :M].double().T @ B[:, :N].double(), pad4(M), pad4(N)

    A, B, r1, lda, ldb = operands()
    A2, B2, r2, _, _ = operands()
    bias = _rand(g, N)
    Cm = torch.full((M, pad4(N)), 7.0, device="cuda")
    assert lib.mtts_gemm_f32_dual(form, M, N, K, P(A), lda, P(B), ldb, P(A2), P(B2), P(Cm), pad4(N), P(bias), 0.5, 0, tile, None) == 0
    torch.cuda.synchronize()
    want = 0.5 * (r1 + r2) + bias.double()[None, :]
    err = (Cm[:, :N].double() - want).abs().max().item()
    assert err < 3e-5 * max(1.0, want.abs().max().item()), err
    assert torch.all(Cm[:, N:] == 7.0)
    # against the two-launch form it replaces (plain, then accumulate): same value up to the order of summation
    C2 = torch.zeros((M, pad4(N)), device="cuda")
    assert lib.mtts_gemm_f32(form, M, N, K, P(A), lda, P(B), ldb, P(C2), pad4(N), P(bias), 0.5, 0, tile, None) == 0
    assert lib.mtts_gemm_f32(form, M, N, K, P(A2), lda, P(B2), ldb, P(C2), pad4(N), None, 0.5, 2, tile, None) == 0
    torch.cuda.synchronize()
    assert (C2[:, :N] - Cm[:, :N]).abs().max().item() < 3e-5 * max(1.0, want.abs().max().item())


@pytest.mark.parametrize("tile", [0, 64, 128, 1064, 1128, 3064, 3128, 4064])
@pytest.mark.parametrize("L,Cin,Cout,k", [(97, 32, 48, 3), (300, 256, 1024, 9), (211, 80, 512, 5), (150, 1024, 256, 1), (64, 512, 80, 5)])
def test_conv1d_fwd_dgrad_wgrad(lib, tile, L, Cin, Cout, k):
    g = np.random.RandomState(L + Cin + Cout + k)
    pad = 4
    x = torch.zeros(L + 2 * pad, Cin, device="cuda"); x[pad:pad + L] = _rand(g, L, Cin)
    w = _rand(g, Cout, Cin, k) / np.sqrt(Cin * k)
    b = _rand(g, Cout)
    wi = w.permute(0, 2, 1).contiguous()  # internal [Cout][k][Cin]
    xt = x[pad:pad + L].T.unsqueeze(0).clone().requires_grad_(True)
    wt = w.clone().requires_grad_(True)
    y_ref = torch.nn.functional.conv1d(xt, wt, b, padding=k // 2)
    dy = torch.zeros(L + 2 * pad, Cout, device="cuda"); dy[pad:pad + L] = _rand(g, L, Cout)
    y_ref.backward(dy[pad:pad + L].T.unsqueeze(0))
    y = torch.zeros(L, Cout, device="cuda")
    assert lib.mtts_conv1d_f32(0, L, Cin, Cout, k, P(x[pad:]), P(wi), P(y), P(b), tile, None) == 0
    dx = torch.zeros(L, Cin, device="cuda")
    assert lib.mtts_conv1d_f32(1, L, Cin, Cout, k, P(dy[pad:]), P(wi), P(dx), None, tile, None) == 0
    dw = torch.zeros(Cout, k, Cin, device="cuda")
    assert lib.mtts_conv1d_f32(2, L, Cin, Cout, k, P(dy[pad:]), P(x[pad:]), P(dw), None, tile, None) == 0
    torch.cuda.synchronize()
    tol = lambda r: 3e-5 * max(1.0, r.abs().max().item())
    yr = y_ref[0].T
    assert (y - yr).abs().max().item() < tol(yr)
    dxr = xt.grad[0].T
    assert (dx - dxr).abs().max().item() < tol(dxr)
    dwr = wt.grad.permute(0, 2, 1)
    assert (dw - dwr).abs().max().item() < tol(dwr) * 4


@pytest.mark.gpu
def test_kloop_variants_are_bit_identical():
    """MTTS_KLOOP = 0 / 1 / 4 (csrc/gemm.h: gemm_f32_kloop KL — the round 2-4 loop, rotating half fragments, the interleaved default) compute
    every accumulator chain in the same k order: NT / NN / TN products with partial last slices and ragged M / N, BK = 32 and the k = 9
    convolution's three forms must give the same bits (tools/kloop_forms.py hash; the switch is read once per process)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    digests = {}
    for kl in ("0", "1", "4"):
        env = dict(os.environ, MTTS_KLOOP=kl)
        r = subprocess.run([sys.executable, os.path.join(root, "tools", "kloop_forms.py"), "hash"], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-1500:]
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("KLOOP")][-1]
        digests[kl] = line.split()[-1]
    assert len(set(digests.values())) == 1, digests
