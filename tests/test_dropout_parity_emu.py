"""Dropout-ON parity of the product's kernel + orchestration sources (SIMT emulator, CPU) against the oracle running the SAME
counter-based masks (oracle/dropout_masks.py): the configuration the reference trains in and bench.py times
(SubLayers.py:54,90; modules.py:223,235; Layers.py:133-134).  The full-size GPU legs are tests/test_gpu_dropout_parity.py."""
import numpy as np
import pytest
import torch

import __graft_entry__ as ge
from oracle_util import O, heads, synth, tiny_dims, torch_buffers, torch_params
from oracle.dropout_masks import DropoutMasks, keep_mask, plan_seed, site_seed
from meta_tts_amd.engine import Engine

MODS = ["speaker_emb", "variance_adaptor", "decoder", "mel_linear", "postnet"]
PROBS = dict(enc=0.2, dec=0.2, vp=0.5, postnet=0.5)


@pytest.fixture(scope="module")
def emu_lib():
    return ge.build_emulator()


def _kw(dims):
    return dict(n_mel=dims.n_mel, vocab=dims.vocab, s_range=(5, 13), d_range=(1, 6), first_len=12)


def _engine(dims, emu_lib, tasks=2, mods=MODS):
    eng = Engine(dims, adapt_modules=mods, max_tasks=tasks, max_B=3, max_S=16, max_T=96, lib_path=emu_lib)
    eng.load_params(synth.make_params(dims, 0))
    return eng


def test_mask_function_statistics_and_fields():
    """keep-rate, independence of the 4 fields of one hash, sensitivity to every argument."""
    rows = np.arange(4, 300)
    m = keep_mask(site_seed(plan_seed(7, 1), 65), 3, rows, 256, 0.2)
    assert abs(m.mean() - 0.8) < 0.01
    for a in range(4):
        assert abs(m[:, a::4].mean() - 0.8) < 0.02
    assert (m != keep_mask(site_seed(plan_seed(7, 2), 65), 3, rows, 256, 0.2)).mean() > 0.2
    assert (m != keep_mask(site_seed(plan_seed(7, 1), 64), 3, rows, 256, 0.2)).mean() > 0.2
    assert (m != keep_mask(site_seed(plan_seed(7, 1), 65), 2, rows, 256, 0.2)).mean() > 0.2
    assert plan_seed(7, 1) != plan_seed(7, 2) != plan_seed(8, 1)


@pytest.mark.parametrize("levels", [("phoneme_level", "phoneme_level"), ("frame_level", "frame_level")])
def test_forward_loss_backward_with_dropout_vs_oracle(emu_lib, levels):
    """Two ragged tasks in one launch group, dropout on: outputs, 6 losses and EVERY parameter gradient against autograd through
    the oracle with the engine's masks.  Covers all five site kinds incl. the folds (dropout inside layernorm_fwd/bwd,
    bn_apply, the PostNet dY readers) and, frame-level, the predictors on the R rectangle."""
    dims = tiny_dims(pitch_level=levels[0], energy_level=levels[1])
    eng = _engine(dims, emu_lib)
    b0 = synth.make_batch(3, 3, speaker=2, pitch_level=levels[0], energy_level=levels[1], **_kw(dims)) if levels[0] == "frame_level" \
        else synth.make_batch(3, 3, speaker=2, **_kw(dims))
    b1 = synth.make_batch(4, 2, speaker=5, pitch_level=levels[0], energy_level=levels[1], **_kw(dims)) if levels[0] == "frame_level" \
        else synth.make_batch(4, 2, speaker=5, **_kw(dims))
    eng.set_batches(0, [b0, b1])
    eng.set_dropout(True, 11)
    eng.forward(0, use_fast=False, train=True)
    dev_loss = eng.loss(0)
    eng.backward(0, use_fast=False, scale=1.0, need_encoder=True)
    for ti, b in enumerate([b0, b1]):
        p = torch_params(dims, requires_grad=True)
        tb = O.to_torch_batch(b)
        dm = DropoutMasks(plan_seed(11, 1), ti, PROBS)
        o = O.fs2_forward(p, torch_buffers(dims), *tb[2:], n_head=heads(dims), max_seq_len=dims.max_seq_len, training=True,
                          pitch_level=levels[0], energy_level=levels[1], dropout=dm)
        lo = O.fs2_loss(tb, o, levels[0], levels[1])
        out = eng.outputs(0, ti)
        for k, ref in (("mel", o[0]), ("mel_post", o[1]), ("p", o[2]), ("e", o[3]), ("logd", o[4])):
            assert np.abs(out[k] - ref.detach().numpy()).max() < 1e-4, (ti, k)
        np.testing.assert_allclose(dev_loss[ti], [float(x) for x in lo], rtol=2e-5)
        names = list(eng.params)
        gs = torch.autograd.grad(lo[0], [p[n] for n in names], allow_unused=True)
        for n, g in zip(names, gs):
            ref = g.numpy() if g is not None else np.zeros(eng.params[n][0], np.float32)
            got = eng.export(n, 2, ti)
            assert np.abs(got - ref).max() <= 1e-3 * np.abs(ref).max() + 2e-7, (ti, n)
    # the dropout-off oracle is NOT what the engine computed (the masks matter at this tolerance)
    p = torch_params(dims)
    with torch.no_grad():
        o = O.fs2_forward(p, torch_buffers(dims), *O.to_torch_batch(b0)[2:], n_head=heads(dims), max_seq_len=dims.max_seq_len, training=True,
                          pitch_level=levels[0], energy_level=levels[1])
    assert np.abs(eng.outputs(0, 0)["mel_post"] - o[1].numpy()).max() > 1e-2
    eng.close()


@pytest.mark.parametrize("second_order", [False, True])
def test_maml_meta_gradient_with_dropout_vs_oracle(emu_lib, second_order):
    """3 inner steps + query pass with dropout on, two tasks grouped: per-step support losses, query 6-tuple and the outer
    gradient of every tensor against O.maml_task given the plan seeds in the order the engine draws them (steps, then the query
    pass) — first order and second order (the Hessian-vector passes replay the inner steps' masks)."""
    dims = tiny_dims()
    eng = _engine(dims, emu_lib)
    kw = _kw(dims)
    tasks = [(synth.make_batch(31 + 2 * j, 3, speaker=2 + j, **kw), synth.make_batch(32 + 2 * j, 2, speaker=2 + j, **kw)) for j in range(2)]
    sup, qry = [t[0] for t in tasks], [t[1] for t in tasks]
    eng.set_batches(0, sup)
    eng.set_batches(1, qry, spk_from=sup, average_spk=True)
    steps, lr, seed = 3, 1e-4, 5
    eng.set_dropout(True, seed)
    q, s = eng.meta_grad(steps, lr, 1.0, second_order=second_order)
    for j, (sb, qb) in enumerate(tasks):
        p = torch_params(dims, requires_grad=True)
        dms = [DropoutMasks(plan_seed(seed, k + 1), j, PROBS) for k in range(steps + 1)]
        ql, sl, _, _ = O.maml_task(p, torch_buffers(dims), O.to_torch_batch(sb), O.to_torch_batch(qb), steps=steps, lr=lr,
                                   second_order=second_order, modules=MODS, n_head=heads(dims), max_seq_len=dims.max_seq_len, dropout=dms)
        np.testing.assert_allclose(q[j], [float(x) for x in ql], rtol=5e-5, err_msg=f"query losses of task {j}")
        np.testing.assert_allclose(s[:, j, :], np.array([[float(x) for x in l] for l in sl]), rtol=5e-5)
        names = list(eng.params)
        gs = torch.autograd.grad(ql[0], [p[n] for n in names], allow_unused=True)
        for n, g in zip(names, gs):
            ref = g.numpy() if g is not None else np.zeros(eng.params[n][0], np.float32)
            got = eng.export(n, 2, j)
            assert np.abs(got - ref).max() <= 2e-3 * np.abs(ref).max() + 1e-6, (j, n, second_order)   # (w_ks.bias: exactly 0 in exact arithmetic)
    eng.close()
