"""Kernel-level C-ABI entry points of the HBM-bound kernels (include/mtts.h: mtts_layernorm_*, mtts_softmax_*, mtts_sdpa_fwd,
mtts_batchnorm_*, mtts_table_grad) against torch on the same seeded inputs.  The same checks run twice: through the SIMT
emulator build on CPU (index math, masks, formulas) and, marked gpu, through libmtts.so on the MI355X (DPP reductions, MFMA
attention products).  Reference ops: transformer/SubLayers.py:55,91 (LayerNorm), Modules.py:14-25 (SDPA),
Layers.py:129-137 (PostNet BatchNorm + tanh), nn.Embedding backward (modules.py:73-78, Models.py:56-58)."""
import ctypes as C

import numpy as np
import pytest
import torch

import __graft_entry__ as ge
from meta_tts_amd import _lib


class Dev:
    """Device-or-host buffers behind one interface: torch CUDA tensors on the GPU, numpy arrays for the emulator (whose 'device'
    memory is host memory)."""

    def __init__(self, gpu):
        self.gpu = gpu
        self.lib = _lib.load(None if gpu else ge.build_emulator())
        self.keep = []

    def put(self, a):
        a = np.ascontiguousarray(a)
        if self.gpu:
            t = torch.from_numpy(a).cuda()
            self.keep.append(t)
            return t
        self.keep.append(a)
        return a

    def empty(self, shape, dtype=np.float32, fill=0):
        return self.put(np.full(shape, fill, dtype))

    def ptr(self, x):
        if x is None:
            return None
        return C.c_void_p(x.data_ptr()) if self.gpu else x.ctypes.data_as(C.c_void_p)

    def get(self, x):
        if self.gpu:
            torch.cuda.synchronize()
            return x.cpu().numpy()
        return x

    def ws(self, rows, n_mat=0):
        return self.empty((int(self.lib.mtts_kernel_ws_bytes(rows, n_mat)) + 3) // 4, np.float32)


@pytest.fixture(params=[pytest.param(False, id="emu"), pytest.param(True, id="gpu", marks=pytest.mark.gpu)])
def dev(request):
    if request.param:
        ge.build_device()
    return Dev(request.param)


@pytest.mark.parametrize("rows,Cc,with_res,with_mask", [(37, 256, True, True), (8, 32, False, False), (203, 1024, True, False), (5, 48, False, True)])
def test_layernorm_fwd_bwd(dev, rows, Cc, with_res, with_mask):
    g = np.random.RandomState(rows + Cc)
    a, res = g.standard_normal((rows, Cc)).astype(np.float32), g.standard_normal((rows, Cc)).astype(np.float32)
    gamma, beta = (1 + 0.1 * g.standard_normal(Cc)).astype(np.float32), (0.1 * g.standard_normal(Cc)).astype(np.float32)
    mask = (g.rand(rows) > 0.3).astype(np.uint8)
    dy = g.standard_normal((rows, Cc)).astype(np.float32)
    d = {k: dev.put(v) for k, v in dict(a=a, res=res, gamma=gamma, beta=beta, mask=mask, dy=dy).items()}
    z, y, st = dev.empty((rows, Cc)), dev.empty((rows, Cc)), dev.empty((rows, 2))
    dz, dgm, dbt = dev.empty((rows, Cc)), dev.empty(Cc), dev.empty(Cc)
    ws = dev.ws(rows)
    P = dev.ptr
    assert dev.lib.mtts_layernorm_fwd(rows, Cc, P(d["a"]), P(d["res"]) if with_res else None, P(d["gamma"]), P(d["beta"]),
                                      P(d["mask"]) if with_mask else None, P(z), P(y), P(st), P(ws), None) == 0
    assert dev.lib.mtts_layernorm_bwd(rows, Cc, P(d["dy"]), P(z), P(st), P(d["gamma"]), P(d["mask"]) if with_mask else None, P(dz), P(dgm), P(dbt),
                                      P(ws), None) == 0
    ta = torch.from_numpy(a).requires_grad_(True)
    tg, tb = torch.from_numpy(gamma).requires_grad_(True), torch.from_numpy(beta).requires_grad_(True)
    zin = ta + (torch.from_numpy(res) if with_res else 0)
    ref = torch.nn.functional.layer_norm(zin, (Cc,), tg, tb, 1e-5)
    m = torch.from_numpy(mask.astype(np.float32))[:, None] if with_mask else torch.ones(rows, 1)
    ref = ref * m
    ref.backward(torch.from_numpy(dy))
    np.testing.assert_allclose(dev.get(y), ref.detach().numpy(), atol=3e-6)
    np.testing.assert_allclose(dev.get(z), zin.detach().numpy(), atol=0)
    np.testing.assert_allclose(dev.get(dz), ta.grad.numpy(), atol=2e-5)
    np.testing.assert_allclose(dev.get(dgm), tg.grad.numpy(), rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(dev.get(dbt), tb.grad.numpy(), rtol=1e-4, atol=2e-5)


@pytest.mark.parametrize("n_mat,L,dk", [(3, 17, 16), (2, 80, 128), (4, 5, 32), (2, 130, 64), (1, 333, 128), (1, 700, 128)])
def test_sdpa_and_softmax(dev, n_mat, L, dk):
    # (the fused attention forward, csrc/attention.h: L <= 128 / <= 640 / <= 1024 take the three LDS tile classes)
    g = np.random.RandomState(L * 7 + dk)
    q, k, v = (g.standard_normal((n_mat, L, dk)).astype(np.float32) for _ in range(3))
    ldS = (L + 3) & ~3
    dq, dk_, dv = dev.put(q), dev.put(k), dev.put(v)
    Pm, o = dev.empty((n_mat, L, ldS), fill=7), dev.empty((n_mat, L, dk))
    ws = dev.ws(1, n_mat)
    P = dev.ptr
    assert dev.lib.mtts_sdpa_fwd(n_mat, L, dk, P(dq), P(dk_), P(dv), P(Pm), P(o), P(ws), None) == 0
    tq, tk, tv = (torch.from_numpy(x) for x in (q, k, v))
    attn = torch.softmax(torch.bmm(tq, tk.transpose(1, 2)) / np.sqrt(dk), dim=2)
    ref = torch.bmm(attn, tv)
    np.testing.assert_allclose(dev.get(Pm)[:, :, :L], attn.numpy(), atol=3e-6)
    np.testing.assert_allclose(dev.get(o), ref.numpy(), atol=2e-5)
    # softmax alone + its backward: dS = alpha * P o (dP - rowsum(dP o P))
    S = np.zeros((n_mat, L, ldS), np.float32)
    S[:, :, :L] = g.standard_normal((n_mat, L, L)).astype(np.float32)
    dS = dev.put(S.copy())
    assert dev.lib.mtts_softmax_fwd(n_mat, L, P(dS), P(ws), None) == 0
    ts = torch.from_numpy(S[:, :, :L].copy()).requires_grad_(True)
    pr = torch.softmax(ts * 0.25, dim=2)  # the scores were scaled by alpha = 0.25 before the softmax: d/dS carries alpha
    got = dev.get(dS)[:, :, :L]
    np.testing.assert_allclose(got, torch.softmax(ts.detach(), dim=2).numpy(), atol=3e-6)
    up = g.standard_normal((n_mat, L, L)).astype(np.float32)
    pr.backward(torch.from_numpy(up))
    Pd = np.zeros((n_mat, L, ldS), np.float32); Pd[:, :, :L] = pr.detach().numpy()
    dP = np.zeros((n_mat, L, ldS), np.float32); dP[:, :, :L] = up
    dPd = dev.put(dP)
    assert dev.lib.mtts_softmax_bwd(n_mat, L, P(dev.put(Pd)), P(dPd), 0.25, P(ws), None) == 0
    np.testing.assert_allclose(dev.get(dPd)[:, :, :L], ts.grad.numpy(), atol=3e-6)


@pytest.mark.parametrize("B,T,Cc,do_tanh", [(3, 21, 48, 1), (2, 40, 512, 1), (1, 9, 80, 0)])
def test_batchnorm_fwd_bwd_over_padded_rectangle(dev, B, T, Cc, do_tanh):
    """rows = G + B * (T + G) with 4 guard rows between sequences (engine.h row space R); the statistics run over all B * T in-rect
    rows, padded frames included, as nn.BatchNorm1d over (B, C, T_max) does in the reference."""
    G = 4
    rows = G + B * (T + G)
    g = np.random.RandomState(B * 100 + T)
    inrect = np.zeros(rows, np.uint8)
    for b in range(B):
        inrect[G + b * (T + G): G + b * (T + G) + T] = 1
    x = g.standard_normal((rows, Cc)).astype(np.float32) * inrect[:, None]
    gamma, beta = (1 + 0.1 * g.standard_normal(Cc)).astype(np.float32), (0.1 * g.standard_normal(Cc)).astype(np.float32)
    dy = g.standard_normal((rows, Cc)).astype(np.float32) * inrect[:, None]
    dx_, dg_, db_, st, y, dxo = dev.put(x), dev.put(gamma), dev.put(beta), dev.empty(3 * Cc), dev.empty((rows, Cc)), dev.empty((rows, Cc))
    dgm, dbt = dev.empty(Cc), dev.empty(Cc)
    ws = dev.ws(rows)
    P = dev.ptr
    assert dev.lib.mtts_batchnorm_fwd(rows, Cc, P(dx_), P(dev.put(inrect)), P(dg_), P(db_), do_tanh, P(st), P(y), P(ws), None) == 0
    assert dev.lib.mtts_batchnorm_bwd(rows, B * T, Cc, P(dev.put(dy)), P(y), P(dx_), P(st), P(dev.put(inrect)), P(dg_), do_tanh, P(dxo), P(dgm), P(dbt),
                                      P(ws), None) == 0
    sel = inrect.astype(bool)
    tx = torch.from_numpy(x[sel]).requires_grad_(True)
    tg, tb = torch.from_numpy(gamma).requires_grad_(True), torch.from_numpy(beta).requires_grad_(True)
    rm, rv = torch.zeros(Cc), torch.ones(Cc)
    ref = torch.nn.functional.batch_norm(tx, rm, rv, tg, tb, True, 1.0, 1e-5)   # momentum 1: running stats := batch stats
    if do_tanh:
        ref = torch.tanh(ref)
    ref.backward(torch.from_numpy(dy[sel]))
    got = dev.get(y)
    np.testing.assert_allclose(got[sel], ref.detach().numpy(), atol=5e-6)
    assert np.all(got[~sel] == 0)
    s = dev.get(st)
    np.testing.assert_allclose(s[:Cc], rm.numpy(), atol=2e-6)
    np.testing.assert_allclose(s[2 * Cc:], rv.numpy(), rtol=2e-5)   # unbiased variance (what the running-var update uses)
    np.testing.assert_allclose(dev.get(dxo)[sel], tx.grad.numpy(), atol=2e-5)
    np.testing.assert_allclose(dev.get(dgm), tg.grad.numpy(), rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(dev.get(dbt), tb.grad.numpy(), rtol=1e-4, atol=2e-5)


@pytest.mark.parametrize("rows,Cc,V,skip", [(150, 32, 20, 0), (300, 256, 361, 0), (64, 16, 7, -1)])
def test_table_grad_matches_embedding_backward(dev, rows, Cc, V, skip):
    g = np.random.RandomState(rows + V)
    idx = g.randint(0, V, rows).astype(np.int32)
    dx = g.standard_normal((rows, Cc)).astype(np.float32)
    dt = dev.empty((V, Cc), fill=9)
    ws = dev.ws(rows)
    P = dev.ptr
    assert dev.lib.mtts_table_grad(rows, Cc, V, P(dev.put(dx)), P(dev.put(idx)), skip, P(dt), P(ws), None) == 0
    w = torch.zeros(V, Cc, requires_grad=True)
    out = torch.nn.functional.embedding(torch.from_numpy(idx.astype(np.int64)), w, padding_idx=skip if skip >= 0 else None)
    out.backward(torch.from_numpy(dx))
    np.testing.assert_allclose(dev.get(dt), w.grad.numpy(), atol=2e-5)
    if skip >= 0:
        assert np.all(dev.get(dt)[skip] == 0)


def test_bad_arguments_are_rejected(dev):
    ws = dev.ws(8)
    assert dev.lib.mtts_layernorm_fwd(8, 30, None, None, None, None, None, None, None, None, dev.ptr(ws), None) != 0  # C % 4, null pointers
    assert dev.lib.mtts_softmax_fwd(0, 4, None, dev.ptr(ws), None) != 0
    assert dev.lib.mtts_kernel_ws_bytes(100, 4) > dev.lib.mtts_kernel_ws_bytes(10, 0) > 0


@pytest.mark.parametrize("S,Cc,with_extras", [(13, 256, True), (40, 32, False)])
def test_length_regulate_fwd_bwd(dev, S, Cc, with_extras):
    """modules.py:167-190: expand every phoneme row by its duration (zero durations drop the phoneme), and the transpose."""
    g = np.random.RandomState(S + Cc)
    dur = g.randint(0, 6, size=S)
    dur[1] = 0
    x = g.standard_normal((S, Cc)).astype(np.float32)
    src = np.repeat(np.arange(S), dur).astype(np.int32)
    T = len(src)
    src = np.concatenate([src, -np.ones(3, np.int32)])          # three padded frame rows
    nf = len(src)
    row_t = np.concatenate([np.arange(T), np.zeros(3)]).astype(np.int32)
    spk = g.standard_normal(Cc).astype(np.float32)
    pos = g.standard_normal((nf + 1, Cc)).astype(np.float32)
    out = dev.empty((nf, Cc), fill=7)
    d = {k: dev.put(v) for k, v in dict(x=x, src=src, row_t=row_t, spk=spk, pos=pos).items()}
    ws = dev.ws(max(nf, S))
    P = dev.ptr
    assert dev.lib.mtts_length_regulate_fwd(nf, Cc, P(d["x"]), P(d["src"]), P(d["spk"]) if with_extras else None, P(d["pos"]) if with_extras else None,
                                            P(d["row_t"]) if with_extras else None, P(out), P(ws), None) == 0
    ref = np.zeros((nf, Cc), np.float32)
    ref[:T] = x[src[:T]] + ((spk[None] + pos[row_t[:T]]) if with_extras else 0)
    np.testing.assert_allclose(dev.get(out), ref, rtol=1e-6, atol=1e-6)
    # transpose: dx[p] = sum of dout over the phoneme's frames (torch: index_add of the gather)
    dout = g.standard_normal((nf, Cc)).astype(np.float32)
    first = np.concatenate([[0], np.cumsum(dur)[:-1]]).astype(np.int32)
    dx = dev.empty((S, Cc), fill=3)
    dd = dev.put(dout)
    assert dev.lib.mtts_length_regulate_bwd(S, Cc, P(dd), P(dev.put(first)), P(dev.put(dur.astype(np.int32))), P(dx), 0, P(ws), None) == 0
    tx = torch.from_numpy(x).requires_grad_(True)
    tx[torch.from_numpy(src[:T].astype(np.int64))].backward(torch.from_numpy(dout[:T]))
    np.testing.assert_allclose(dev.get(dx), tx.grad.numpy(), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("rows,Cc,full", [(19, 256, True), (6, 64, False)])
def test_layernorm_and_softmax_jvp(dev, rows, Cc, full):
    """Tangent twins (csrc/tangent.h) against torch.func.jvp of layer_norm / softmax."""
    g = np.random.RandomState(rows * 3 + Cc)
    a, ta = (g.standard_normal((rows, Cc)).astype(np.float32) for _ in range(2))
    tres = g.standard_normal((rows, Cc)).astype(np.float32)
    gamma, beta = (1 + 0.1 * g.standard_normal(Cc)).astype(np.float32), (0.1 * g.standard_normal(Cc)).astype(np.float32)
    tgamma, tbeta = (g.standard_normal(Cc).astype(np.float32) for _ in range(2))
    mask = (g.rand(rows) > 0.3).astype(np.uint8)
    d = {k: dev.put(v) for k, v in dict(a=a, ta=ta, tres=tres, gamma=gamma, beta=beta, tgamma=tgamma, tbeta=tbeta, mask=mask).items()}
    z, y, st, ty = dev.empty((rows, Cc)), dev.empty((rows, Cc)), dev.empty((rows, 2)), dev.empty((rows, Cc), fill=5)
    ws = dev.ws(rows)
    P = dev.ptr
    assert dev.lib.mtts_layernorm_fwd(rows, Cc, P(d["a"]), None, P(d["gamma"]), P(d["beta"]), P(d["mask"]) if full else None, P(z), P(y), P(st), P(ws), None) == 0
    assert dev.lib.mtts_layernorm_jvp(rows, Cc, P(d["ta"]), P(d["tres"]) if full else None, P(z), P(st), P(d["gamma"]), P(d["tgamma"]) if full else None,
                                      P(d["tbeta"]) if full else None, P(d["mask"]) if full else None, P(ty), P(ws), None) == 0
    m = torch.from_numpy(mask.astype(np.float32))[:, None] if full else torch.ones(rows, 1)
    f = lambda x, gm, bt: torch.nn.functional.layer_norm(x, (Cc,), gm, bt, 1e-5) * m
    tin = torch.from_numpy(ta + (tres if full else 0))
    tg = torch.from_numpy(tgamma) if full else torch.zeros(Cc)
    tb = torch.from_numpy(tbeta) if full else torch.zeros(Cc)
    _, ref = torch.func.jvp(f, (torch.from_numpy(a), torch.from_numpy(gamma), torch.from_numpy(beta)), (tin, tg, tb))
    np.testing.assert_allclose(dev.get(ty), ref.numpy(), rtol=2e-4, atol=2e-5)
    # softmax
    n_mat, L = 3, rows
    ldS = (L + 3) & ~3
    S0, tS0 = (g.standard_normal((n_mat, L, L)).astype(np.float32) for _ in range(2))
    Sp, tSp = np.zeros((n_mat, L, ldS), np.float32), np.zeros((n_mat, L, ldS), np.float32)
    Sp[:, :, :L], tSp[:, :, :L] = S0, tS0
    dS, dtS = dev.put(Sp), dev.put(tSp)
    ws2 = dev.ws(1, n_mat)
    assert dev.lib.mtts_softmax_fwd(n_mat, L, P(dS), P(ws2), None) == 0
    assert dev.lib.mtts_softmax_jvp(n_mat, L, P(dS), P(dtS), P(ws2), None) == 0
    _, ref = torch.func.jvp(lambda s: torch.softmax(s, -1), (torch.from_numpy(S0),), (torch.from_numpy(tS0),))
    np.testing.assert_allclose(dev.get(dtS)[:, :, :L], ref.numpy(), rtol=2e-4, atol=2e-6)


@pytest.mark.parametrize("tile", [0, 64, 4064])
@pytest.mark.parametrize("form", [0, 1, 2])
@pytest.mark.parametrize("M,N,K", [(70, 40, 36), (33, 130, 100), (129, 64, 530)])
def test_gemm_dual_source(dev, form, tile, M, N, K):
    """mtts_gemm_f32_dual: C = alpha (A B + A2 B2) + bias in ONE accumulator chain (csrc/gemm.h: GemmArgs::A2 — the shape of every tangent
    product of second-order MAML, base_adaptor.py:107) against float64, for the three operand forms, through the launch queue (tile 0),
    a plain 64x64 grid and the LDS-DMA family."""
    g = np.random.RandomState(M + 3 * N + 7 * K + form)
    pad4 = lambda x: (x + 3) & ~3

    def operands():
        if form == 0:
            A, B = np.zeros((M, pad4(K)), np.float32), np.zeros((N, pad4(K)), np.float32)
            A[:, :K], B[:, :K] = g.standard_normal((M, K)), g.standard_normal((N, K))
            return A, B, A[:, :K].astype(np.float64) @ B[:, :K].astype(np.float64).T, pad4(K), pad4(K)
        if form == 1:
            A, B = np.zeros((M, pad4(K)), np.float32), g.standard_normal((K, pad4(N))).astype(np.float32)
            A[:, :K] = g.standard_normal((M, K))
            return A, B, A[:, :K].astype(np.float64) @ B[:, :N].astype(np.float64), pad4(K), pad4(N)
        A, B = g.standard_normal((K, pad4(M))).astype(np.float32), g.standard_normal((K, pad4(N))).astype(np.float32)
        return A, B, A[:, :M].astype(np.float64).T @ B[:, :N].astype(np.float64), pad4(M), pad4(N)

    A, B, r1, lda, ldb = operands()
    A2, B2, r2, _, _ = operands()
    bias = g.standard_normal(N).astype(np.float32)
    dA, dB, dA2, dB2, dbias = (dev.put(x) for x in (A, B, A2, B2, bias))
    out = dev.empty((M, pad4(N)), fill=7.0)
    P = dev.ptr
    assert dev.lib.mtts_gemm_f32_dual(form, M, N, K, P(dA), lda, P(dB), ldb, P(dA2), P(dB2), P(out), pad4(N), P(dbias), 0.5, 0, tile, None) == 0
    got = dev.get(out)
    want = 0.5 * (r1 + r2) + bias[None, :].astype(np.float64)
    assert np.abs(got[:, :N] - want).max() < 3e-5 * max(1.0, np.abs(want).max())
    assert np.all(got[:, N:] == 7.0)
    # a missing second source is an error, not a silent single product
    assert dev.lib.mtts_gemm_f32_dual(form, M, N, K, P(dA), lda, P(dB), ldb, None, None, P(out), pad4(N), None, 1.0, 0, tile, None) != 0
