"""Dropout-ON parity against the REFERENCE ITSELF (tests/golden/make_dropout_golden.py: the reference model in train mode with only the
Bernoulli draw of its nn.Dropout modules / F.dropout calls replaced by oracle/dropout_masks.py masks).  Pins where the five dropout
site kinds sit, on which tensor / layout and with which p and scaling (transformer/SubLayers.py:54,90; lightning/model/modules.py:223,235;
transformer/Layers.py:133-134) for (a) the oracle's dropout mode — CPU — and (b) the engine through the C ABI — `-m gpu`."""
import os

import numpy as np
import pytest
import torch

from oracle_util import O, SMALL, heads, synth, torch_buffers, torch_params
from oracle.dropout_masks import DropoutMasks, plan_seed
from meta_tts_amd.config import ModelDims, default_algorithm_config

DIMS = ModelDims()
MODS = default_algorithm_config()["adapt"]["modules"]
TOL = 2e-5


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def test_fixture_records_the_reference_sites(golden_dir):
    """What the reference constructed and called (recorded by the generator, not assumed here): 8 + 12 + 6 nn.Dropout modules with the
    config's p (encoder / decoder 0.2, predictors 0.5), and per forward 31 dropout calls in the order encoder, predictors (duration, pitch,
    energy), decoder, PostNet (5 x F.dropout 0.5)."""
    g = _load(golden_dir, "small_grad_dropout.npz")
    sites = dict(zip((int(s) for s in g["module_sites"]), (float(p) for p in g["module_probs"])))
    dm = DropoutMasks(1)
    for l in range(DIMS.enc_layers):
        assert sites[2 * l] == sites[2 * l + 1] == dm.probs["enc"]
    for l in range(DIMS.dec_layers):
        assert sites[64 + 2 * l] == sites[65 + 2 * l] == dm.probs["dec"]
    for b in (128, 132, 136):
        assert sites[b] == sites[b + 1] == dm.probs["vp"]
    calls = [int(s) for s in g["call_sites"]]
    assert calls == list(range(8)) + [128, 129, 132, 133, 136, 137] + list(range(64, 76)) + list(range(192, 197))
    assert all(float(p) == dm.probs["postnet"] for p in g["call_probs"][-5:])


def test_oracle_dropout_mode_small_batch_vs_reference(golden_dir):
    g = _load(golden_dir, "small_grad_dropout.npz")
    seed, task = (int(x) for x in g["seed"])
    p = torch_params(DIMS, requires_grad=True)
    buf = torch_buffers(DIMS)
    b = O.to_torch_batch(synth.make_batch(11, 3, speaker=5, **SMALL))
    o = O.fs2_forward(p, buf, *b[2:], n_head=heads(DIMS), training=True, dropout=DropoutMasks(plan_seed(seed, 1), task))
    lo = O.fs2_loss(b, o)
    np.testing.assert_allclose([float(x) for x in lo], g["losses"], rtol=1e-5)
    for key, val in (("mel", o[0]), ("mel_post", o[1]), ("p", o[2]), ("e", o[3]), ("logd", o[4])):
        assert np.abs(val.detach().numpy() - g[key]).max() < TOL, key
    names = [str(n) for n in g["grad_names"]]
    grads = torch.autograd.grad(lo[0], [p[n] for n in names], allow_unused=True)
    gd = {n: (x if x is not None else torch.zeros_like(p[n])) for n, x in zip(names, grads)}
    norms = np.array([float(gd[n].double().norm()) for n in names])
    np.testing.assert_allclose(norms, g["grad_norms"], rtol=2e-4, atol=1e-7)
    for key in g.files:
        if not key.startswith("grad::"):
            continue
        n = key[len("grad::"):]
        got = gd["speaker_emb.model.weight"][5].numpy() if n == "speaker_row" else gd[n].numpy()
        got = got[:4] if (n != "speaker_row" and got.ndim >= 2) else got
        assert np.abs(got - g[key]).max() <= 1e-4 * max(1e-3, np.abs(g[key]).max()), key
    for i in range(5):   # BatchNorm statistics are taken over the DROPPED activations of the layer before
        np.testing.assert_allclose(buf[f"postnet.convolutions.{i}.1.running_mean"].numpy(), g[f"bn{i}_running_mean"], atol=1e-6)
        np.testing.assert_allclose(buf[f"postnet.convolutions.{i}.1.running_var"].numpy(), g[f"bn{i}_running_var"], rtol=1e-5)
    # and the masks matter: the identity-dropout fixture is a different function
    assert np.abs(g["mel_post"] - _load(golden_dir, "small_grad.npz")["mel_post"]).max() > 1e-2


@pytest.mark.parametrize("order", ["fo", "so"])
def test_oracle_dropout_mode_maml_vs_reference(golden_dir, order):
    g = _load(golden_dir, "maml_small_lr1e-3_scaled_dropout.npz")
    seed, task = (int(x) for x in g["seed"])
    p = torch_params(DIMS, requires_grad=True, weight_scale=0.5)
    sup = O.to_torch_batch(synth.make_batch(21, 3, speaker=9, **SMALL))
    qry = O.to_torch_batch(synth.make_batch(22, 3, speaker=9, **SMALL))
    dms = [DropoutMasks(plan_seed(seed, k + 1), task) for k in range(6)]
    ql, sup_losses, fast, preds = O.maml_task(p, torch_buffers(DIMS), sup, qry, steps=5, lr=0.001, second_order=(order == "so"),
                                              modules=MODS, n_head=heads(DIMS), dropout=dms)
    np.testing.assert_allclose(np.array([[float(x) for x in l] for l in sup_losses]), g[f"{order}_sup_losses"], rtol=2e-5)
    np.testing.assert_allclose([float(x) for x in ql], g[f"{order}_qry_losses"], rtol=2e-5)
    assert np.abs(preds[1].detach().numpy() - g[f"{order}_qry_mel_post"]).max() < 1e-4
    names = [str(n) for n in g[f"{order}_outer_names"]]
    og = torch.autograd.grad(ql[0], [p[n] for n in names], allow_unused=True)
    og = {n: (x if x is not None else torch.zeros_like(p[n])) for n, x in zip(names, og)}
    norms = np.array([float(og[n].double().norm()) for n in names])
    np.testing.assert_allclose(norms, g[f"{order}_outer_norms"], rtol=5e-4, atol=1e-7)
    deltas = np.array([float((fast[k] - p[k]).detach().double().norm()) for k in O.adapted_names(p, MODS)])
    np.testing.assert_allclose(deltas, g[f"{order}_delta_norms"], rtol=5e-4, atol=1e-9)
    for key in g.files:
        if not key.startswith(order + "_grad::"):
            continue
        n = key[len(order + "_grad::"):]
        got = og["speaker_emb.model.weight"][9].numpy() if n == "speaker_row" else og[n].numpy()
        got = got[:4] if (n != "speaker_row" and got.ndim >= 2) else got
        assert np.abs(got - g[key]).max() <= 5e-4 * max(1e-3, np.abs(g[key]).max()), key


# ------------------------------------------------------------------ the engine (HIP path through the C ABI) -----------------------------

def _engine():
    import __graft_entry__ as ge
    from meta_tts_amd.engine import Engine
    ge.build_device()
    eng = Engine(DIMS, adapt_modules=MODS, max_tasks=1, max_B=3, max_S=16, max_T=96)
    return eng


@pytest.mark.gpu
def test_engine_small_batch_dropout_on_vs_reference(golden_dir):
    """Forward, 6 losses, every parameter-gradient norm, sampled gradients and the BatchNorm running buffers of one train-mode pass with
    dropout ON against what the reference's own modules computed with the same masks.  mel L1 <= 1e-4 (north-star gate)."""
    g = _load(golden_dir, "small_grad_dropout.npz")
    seed, task = (int(x) for x in g["seed"])
    assert task == 0
    eng = _engine()
    eng.load_params(synth.make_params(DIMS, 0))
    eng.set_batches(0, [synth.make_batch(11, 3, speaker=5, **SMALL)])
    eng.set_dropout(True, seed)
    eng.forward(0, train=True)
    np.testing.assert_allclose(eng.loss(0)[0], g["losses"], rtol=2e-5)
    out = eng.outputs(0, 0)
    for k in ("mel", "mel_post", "p", "e", "logd"):
        d = np.abs(out[k] - g[k])
        assert d.max() < 3e-4 and d.mean() < 5e-5, (k, d.max(), d.mean())
    assert float(np.abs(out["mel_post"] - g["mel_post"]).mean()) < 1e-4
    eng.backward(0, scale=1.0, need_encoder=True)
    names = [str(n) for n in g["grad_names"]]
    norms = np.array([float(np.linalg.norm(eng.export(n, 2, 0).astype(np.float64))) for n in names])
    np.testing.assert_allclose(norms, g["grad_norms"], rtol=5e-3, atol=2e-6)
    for key in g.files:
        if not key.startswith("grad::"):
            continue
        n = key[len("grad::"):]
        got = eng.export("speaker_emb.model.weight", 2, 0)[5] if n == "speaker_row" else eng.export(n, 2, 0)
        got = got[:4] if (n != "speaker_row" and got.ndim >= 2) else got
        assert np.abs(got - g[key]).max() <= 1e-3 * max(1e-3, np.abs(g[key]).max()), key
    for i in range(5):
        m, v, t = eng.get_bn_buffers(i)
        np.testing.assert_allclose(m, g[f"bn{i}_running_mean"], atol=2e-6)
        np.testing.assert_allclose(v, g[f"bn{i}_running_var"], rtol=1e-4)
    eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize("order", ["fo", "so"])
def test_engine_maml_dropout_on_vs_reference(golden_dir, order):
    """Five inner steps at the reference's inner lr + the query pass, dropout ON in every pass (first order, and second order: the
    Hessian-vector passes replay the inner steps' masks), against the reference model's autograd with the same masks."""
    g = _load(golden_dir, "maml_small_lr1e-3_scaled_dropout.npz")
    seed, task = (int(x) for x in g["seed"])
    sup = synth.make_batch(21, 3, speaker=9, **SMALL)
    qry = synth.make_batch(22, 3, speaker=9, **SMALL)
    eng = _engine()
    eng.load_params(synth.make_params(DIMS, 0, weight_scale=0.5))
    eng.set_batches(0, [sup])
    eng.set_batches(1, [qry], spk_from=[sup], average_spk=True)
    eng.set_dropout(True, seed)
    q, s = eng.meta_grad(5, 0.001, 1.0, second_order=(order == "so"))
    np.testing.assert_allclose(s[:, 0, :], g[f"{order}_sup_losses"], rtol=5e-4)
    np.testing.assert_allclose(q[0], g[f"{order}_qry_losses"], rtol=5e-4)
    names = [str(n) for n in g[f"{order}_outer_names"]]
    norms = np.array([float(np.linalg.norm(eng.export(n, 1).astype(np.float64))) for n in names])
    np.testing.assert_allclose(norms, g[f"{order}_outer_norms"], rtol=1e-2, atol=1e-5)
    if order == "so":
        assert np.abs(norms - g["fo_outer_norms"]).max() > 1e-3  # not the first-order answer
    deltas = np.array([float(np.linalg.norm((eng.export(n, 3, 0) - eng.export(n, 0)).astype(np.float64))) for n in g["adapted_names"]])
    np.testing.assert_allclose(deltas, g[f"{order}_delta_norms"], rtol=5e-3, atol=1e-7)
    for key in g.files:
        if not key.startswith(order + "_grad::"):
            continue
        n = key[len(order + "_grad::"):]
        got = eng.export("speaker_emb.model.weight", 1)[9] if n == "speaker_row" else eng.export(n, 1)
        got = got[:4] if (n != "speaker_row" and got.ndim >= 2) else got
        assert np.abs(got - g[key]).max() <= 1e-2 * max(1e-3, np.abs(g[key]).max()), key
    eng.close()
