"""CPU execution of the *product's own kernel and orchestration sources* through the SIMT emulator
(tests/emu) against the oracle: forward, loss, hand-derived backward, first-order MAML, clip + Adam.
Validates indexing / layouts / masks / gradient formulas without a GPU; the MFMA operand mapping
itself is only exercised by the -m gpu tests."""
import numpy as np
import pytest
import torch

import __graft_entry__ as ge
from oracle_util import O, heads, synth, tiny_dims, torch_buffers, torch_params
from meta_tts_amd.engine import Engine

MODS = ["speaker_emb", "variance_adaptor", "decoder", "mel_linear", "postnet"]


@pytest.fixture(scope="module")
def emu_lib():
    return ge.build_emulator()


def _kw(dims):
    return dict(n_mel=dims.n_mel, vocab=dims.vocab, s_range=(5, 13), d_range=(1, 6), first_len=12)


def _engine(dims, emu_lib, tasks=2, mods=MODS):
    eng = Engine(dims, adapt_modules=mods, max_tasks=tasks, max_B=3, max_S=16, max_T=96, lib_path=emu_lib)
    eng.load_params(synth.make_params(dims, 0))
    return eng


def test_param_roundtrip_and_layout(emu_lib):
    dims = tiny_dims()
    eng = _engine(dims, emu_lib)
    ref = synth.make_params(dims, 0)
    for n, (shape, off, adapted) in eng.params.items():
        np.testing.assert_array_equal(eng.export(n), ref[n])
        assert adapted == (n.split(".")[0] in MODS)
        assert off % 4 == 0
    assert set(eng.params) == {k for k in ref if not k.endswith(("position_enc", "pitch_bins", "energy_bins"))}
    eng.close()


def test_forward_loss_backward_two_ragged_tasks(emu_lib):
    dims = tiny_dims()
    eng = _engine(dims, emu_lib)
    b0 = synth.make_batch(3, 3, speaker=2, **_kw(dims))
    b1 = synth.make_batch(4, 2, speaker=5, **_kw(dims))
    eng.set_batches(0, [b0, b1])
    eng.forward(0, use_fast=False, train=True)
    dev_loss = eng.loss(0)
    eng.backward(0, use_fast=False, scale=1.0, need_encoder=True)
    for ti, b in enumerate([b0, b1]):
        p = torch_params(dims, requires_grad=True)
        tb = O.to_torch_batch(b)
        o = O.fs2_forward(p, torch_buffers(dims), *tb[2:], n_head=heads(dims), max_seq_len=dims.max_seq_len, training=True)
        lo = O.fs2_loss(tb, o)
        out = eng.outputs(0, ti)
        for k, ref in (("mel", o[0]), ("mel_post", o[1]), ("p", o[2]), ("e", o[3]), ("logd", o[4])):
            assert np.abs(out[k] - ref.detach().numpy()).max() < 5e-5, (ti, k)
        np.testing.assert_allclose(dev_loss[ti], [float(x) for x in lo], rtol=2e-5)
        names = list(eng.params)
        gs = torch.autograd.grad(lo[0], [p[n] for n in names], allow_unused=True)
        for n, g in zip(names, gs):
            ref = g.numpy() if g is not None else np.zeros(eng.params[n][0], np.float32)
            got = eng.export(n, 2, ti)
            assert np.abs(got - ref).max() <= 1e-3 * np.abs(ref).max() + 2e-7, (ti, n)
    eng.close()


def test_eval_mode_uses_running_stats(emu_lib):
    dims = tiny_dims()
    eng = _engine(dims, emu_lib, tasks=1)
    b0 = synth.make_batch(5, 2, speaker=1, **_kw(dims))
    eng.set_batches(0, [b0])
    p = torch_params(dims)
    buf = torch_buffers(dims)
    tb = O.to_torch_batch(b0)
    eng.forward(0, train=True)  # updates running stats once
    with torch.no_grad():
        O.fs2_forward(p, buf, *tb[2:], n_head=heads(dims), max_seq_len=dims.max_seq_len, training=True)
    for i in range(dims.postnet_layers):
        m, v, t = eng.get_bn_buffers(i)
        np.testing.assert_allclose(m, buf[f"postnet.convolutions.{i}.1.running_mean"].numpy(), atol=1e-6)
        np.testing.assert_allclose(v, buf[f"postnet.convolutions.{i}.1.running_var"].numpy(), rtol=1e-5)
        assert t == 1
    eng.forward(0, train=False)
    with torch.no_grad():
        o = O.fs2_forward(p, buf, *tb[2:], n_head=heads(dims), max_seq_len=dims.max_seq_len, training=False)
    assert np.abs(eng.outputs(0, 0)["mel_post"] - o[1].numpy()).max() < 5e-5
    eng.close()


def test_decoder_truncation_at_max_seq_len(emu_lib):
    """Models.py:154-162: training truncates frames beyond max_seq_len (64 in the tiny config)."""
    dims = tiny_dims()
    eng = _engine(dims, emu_lib, tasks=1)
    b0 = synth.make_batch(6, 2, speaker=1, n_mel=dims.n_mel, vocab=dims.vocab, s_range=(10, 14), d_range=(5, 9), first_len=13)
    assert b0[8] > dims.max_seq_len
    eng.set_batches(0, [b0])
    eng.forward(0, train=True)
    tb = O.to_torch_batch(b0)
    p = torch_params(dims)
    with torch.no_grad():
        o = O.fs2_forward(p, torch_buffers(dims), *tb[2:], n_head=heads(dims), max_seq_len=dims.max_seq_len, training=True)
        lo = O.fs2_loss(tb, o)
    out = eng.outputs(0, 0)
    assert out["mel_post"].shape == tuple(o[1].shape)
    assert np.abs(out["mel_post"] - o[1].numpy()).max() < 5e-5
    np.testing.assert_allclose(eng.loss(0)[0], [float(x) for x in lo], rtol=2e-5)
    eng.close()


def test_first_order_maml_and_outer_update(emu_lib):
    dims = tiny_dims()
    eng = _engine(dims, emu_lib)
    tasks = [(synth.make_batch(10 + 2 * j, 3, speaker=2 + j, **_kw(dims)), synth.make_batch(11 + 2 * j, 2, speaker=2 + j, **_kw(dims)))
             for j in range(2)]
    eng.set_batches(0, [t[0] for t in tasks])
    eng.set_batches(1, [t[1] for t in tasks], spk_from=[t[0] for t in tasks], average_spk=True)
    lr = 0.01
    q, s = eng.meta_grad(3, lr, 0.5)
    tot = None
    for j, (sup, qry) in enumerate(tasks):
        p = torch_params(dims, requires_grad=True)
        ql, sl, fast, _ = O.maml_task(p, torch_buffers(dims), O.to_torch_batch(sup), O.to_torch_batch(qry), steps=3, lr=lr,
                                      second_order=False, modules=MODS, n_head=heads(dims), max_seq_len=dims.max_seq_len)
        np.testing.assert_allclose(s[:, j, 0], [float(l[0]) for l in sl], rtol=5e-5)
        np.testing.assert_allclose(q[j], [float(x) for x in ql], rtol=5e-5)
        for n in O.adapted_names(p, MODS):
            ref = fast[n].detach().numpy()
            assert np.abs(eng.export(n, 3, j) - ref).max() <= 1e-5 * max(np.abs(ref).max(), 1e-3), n
        names = list(eng.params)
        gs = torch.autograd.grad(ql[0], [p[n] for n in names], allow_unused=True)
        g = {n: (x.numpy() * 0.5 if x is not None else np.zeros(eng.params[n][0], np.float32)) for n, x in zip(names, gs)}
        tot = g if tot is None else {n: tot[n] + g[n] for n in names}
    for n in eng.params:
        assert np.abs(eng.export(n, 1) - tot[n]).max() <= 2e-3 * np.abs(tot[n]).max() + 2e-7, n
    before = eng.state_dict()
    norm = eng.outer_update(lr=1e-3, fetch_norm=True)
    ref_norm = float(np.sqrt(sum((tot[n].astype(np.float64) ** 2).sum() for n in tot)))
    assert abs(norm - ref_norm) < 1e-4 * ref_norm
    after = eng.state_dict()
    coef = min(1.0, 1.0 / (ref_norm + 1e-6))
    for n in eng.params:
        x = torch.from_numpy(before[n].copy()); g = torch.from_numpy(tot[n].copy()) * coef
        O.adam_step(x, g, torch.zeros_like(x), torch.zeros_like(x), 1, 1e-3)
        big = np.abs(tot[n]) * coef > 1e-5  # Adam's first step is lr*sign(g): only meaningful off the noise floor
        assert np.abs(after[n] - x.numpy())[big].max(initial=0.0) < 2e-5, n
    eng.close()


def test_non_default_adapt_modules(emu_lib):
    """adapt.modules is configurable (config/algorithm/*.yaml): a different split moves the fast-weight slice."""
    dims = tiny_dims()
    eng = _engine(dims, emu_lib, tasks=1, mods=["speaker_emb", "decoder"])
    assert all(a == (n.split(".")[0] in ("speaker_emb", "decoder")) for n, (_, _, a) in eng.params.items())
    sup = synth.make_batch(30, 2, speaker=3, **_kw(dims)); qry = synth.make_batch(31, 2, speaker=3, **_kw(dims))
    eng.set_batches(0, [sup]); eng.set_batches(1, [qry], spk_from=[sup], average_spk=True)
    q, _ = eng.meta_grad(2, 0.01, 1.0)
    p = torch_params(dims, requires_grad=True)
    ql, _, _, _ = O.maml_task(p, torch_buffers(dims), O.to_torch_batch(sup), O.to_torch_batch(qry), steps=2, lr=0.01,
                              second_order=False, modules=["speaker_emb", "decoder"], n_head=heads(dims), max_seq_len=dims.max_seq_len)
    np.testing.assert_allclose(q[0], [float(x) for x in ql], rtol=5e-5)
    for n in ("variance_adaptor.pitch_predictor.linear_layer.weight", "decoder.layer_stack.0.pos_ffn.w_1.weight"):
        g = torch.autograd.grad(ql[0], p[n], retain_graph=True)[0].numpy()
        assert np.abs(eng.export(n, 1) - g).max() <= 2e-3 * np.abs(g).max() + 2e-7, n
    eng.close()


def test_free_running_synthesis_matches_oracle(emu_lib):
    """modules.py:132-137 path: predicted durations size the frame spaces (one device->host copy per task);
    p/e controls; a zero-length utterance; eval and train (post-adaptation) modes."""
    dims = tiny_dims()
    params = synth.make_params(dims, 0)
    params["variance_adaptor.duration_predictor.linear_layer.bias"][:] = 1.2  # some non-zero durations at random init
    eng = Engine(dims, adapt_modules=[], max_tasks=2, max_B=3, max_S=16, max_T=96, lib_path=emu_lib)
    eng.load_params(params)
    b0 = synth.make_batch(3, 3, speaker=2, **_kw(dims)); b1 = synth.make_batch(4, 2, speaker=5, **_kw(dims))
    p = {k: torch.from_numpy(v.copy()) for k, v in params.items()}
    for train in (False, True):
        eng.set_batches(0, [b0[:6], b1[:6]])
        eng.synthesize(0, train=train, p_control=1.1, e_control=0.9, d_control=1.0)
        for ti, b in enumerate([b0, b1]):
            tb = O.to_torch_batch(b)
            with torch.no_grad():
                o = O.fs2_forward(p, torch_buffers(dims), *tb[2:6], p_control=1.1, e_control=0.9, n_head=heads(dims),
                                  max_seq_len=dims.max_seq_len, training=train)
            out = eng.outputs(0, ti)
            np.testing.assert_array_equal(out["d_rounded"], o[5].numpy())
            np.testing.assert_array_equal(out["mel_lens"], o[9].numpy())
            assert out["mel_post"].shape == tuple(o[1].shape)
            assert np.abs(out["mel_post"] - o[1].numpy()).max() < 5e-5
    with pytest.raises(Exception):
        eng.loss(0)  # no targets
    eng.close()


ENC_MODS = ["encoder"] + MODS   # config/algorithm/dev.yaml:28-33 adapts the encoder too


def test_hessian_vector_product_matches_double_backward(emu_lib):
    _check_hvp(emu_lib, MODS)


def test_hessian_vector_product_with_adapted_encoder(emu_lib):
    _check_hvp(emu_lib, ENC_MODS)


def _check_hvp(emu_lib, mods):
    """The forward-over-reverse HVP (csrc/engine_so.inc) against torch's create_graph double backward, every tensor,
    two ragged tasks; direction v = the support gradient itself (adapted slice).  With the encoder among the adapted modules the
    tangents start at the word-embedding table instead of at the encoder output."""
    MODS = mods
    dims = tiny_dims()
    eng = _engine(dims, emu_lib, mods=mods)
    b0 = synth.make_batch(3, 3, speaker=2, **_kw(dims)); b1 = synth.make_batch(4, 2, speaker=5, **_kw(dims))
    eng.set_batches(0, [b0, b1])
    eng.adapt(0, 0.0, reset=True)  # fast weights := theta
    eng.forward(0, use_fast=True, train=True)
    eng.backward(0, use_fast=True, scale=1.0, need_encoder=True)
    eng.hvp_support()
    for ti, b in enumerate([b0, b1]):
        p = torch_params(dims, requires_grad=True)
        tb = O.to_torch_batch(b)
        lo = O.fs2_loss(tb, O.fs2_forward(p, torch_buffers(dims), *tb[2:], n_head=heads(dims), max_seq_len=dims.max_seq_len, training=True))
        an = O.adapted_names(p, MODS)
        g = torch.autograd.grad(lo[0], [p[n] for n in an], create_graph=True)
        dot = sum((gi * gi.detach()).sum() for gi in g)
        names = list(eng.params)
        hv = torch.autograd.grad(dot, [p[n] for n in names], allow_unused=True)
        scale = max(float(h.abs().max()) for h in hv if h is not None)
        for n, h in zip(names, hv):
            ref = h.numpy() if h is not None else np.zeros(eng.params[n][0], np.float32)
            got = eng.export(n, 6, ti)
            assert np.abs(got - ref).max() <= 2e-3 * np.abs(ref).max() + 1e-7 * scale, (ti, n)
    eng.close()


def test_second_order_maml_matches_oracle(emu_lib):
    _check_second_order(emu_lib, MODS)


def test_second_order_maml_with_adapted_encoder(emu_lib):
    _check_second_order(emu_lib, ENC_MODS)


def _check_second_order(emu_lib, mods):
    """Training mode of the reference (first_order = not train, base_adaptor.py:107): outer gradient through 3 inner steps."""
    MODS = mods
    dims = tiny_dims()
    eng = _engine(dims, emu_lib, mods=mods)
    tasks = [(synth.make_batch(10 + 2 * j, 3, speaker=2 + j, **_kw(dims)), synth.make_batch(11 + 2 * j, 2, speaker=2 + j, **_kw(dims)))
             for j in range(2)]
    eng.set_batches(0, [t[0] for t in tasks])
    eng.set_batches(1, [t[1] for t in tasks], spk_from=[t[0] for t in tasks], average_spk=True)
    lr = 0.01
    q, s = eng.meta_grad(3, lr, 0.5, second_order=True)
    tot, fo = None, None
    for j, (sup, qry) in enumerate(tasks):
        for so in (True, False):
            p = torch_params(dims, requires_grad=True)
            ql, sl, _, _ = O.maml_task(p, torch_buffers(dims), O.to_torch_batch(sup), O.to_torch_batch(qry), steps=3, lr=lr,
                                       second_order=so, modules=MODS, n_head=heads(dims), max_seq_len=dims.max_seq_len)
            names = list(eng.params)
            gs = torch.autograd.grad(ql[0], [p[n] for n in names], allow_unused=True)
            g = {n: (x.numpy() * 0.5 if x is not None else np.zeros(eng.params[n][0], np.float32)) for n, x in zip(names, gs)}
            if so:
                np.testing.assert_allclose(q[j], [float(x) for x in ql], rtol=5e-5)
                tot = g if tot is None else {n: tot[n] + g[n] for n in names}
            else:
                fo = g if fo is None else {n: fo[n] + g[n] for n in names}
    scale = max(np.abs(v).max() for v in tot.values())
    for n in eng.params:
        assert np.abs(eng.export(n, 1) - tot[n]).max() <= 3e-3 * np.abs(tot[n]).max() + 1e-7 * scale, n
    # and it is genuinely different from the first-order gradient (incl. the non-adapted encoder)
    assert np.abs(fo["encoder.layer_stack.0.pos_ffn.w_1.weight"] - tot["encoder.layer_stack.0.pos_ffn.w_1.weight"]).max() > \
        0.05 * np.abs(tot["encoder.layer_stack.0.pos_ffn.w_1.weight"]).max()
    # first-order call afterwards still gives the first-order answer (indirections restored)
    eng.meta_grad(3, lr, 0.5, second_order=False)
    n = "decoder.layer_stack.1.pos_ffn.w_2.weight"
    assert np.abs(eng.export(n, 1) - fo[n]).max() <= 2e-3 * np.abs(fo[n]).max()
    eng.close()


def _loss_and_grads(eng, seed, names, use_fast=False):
    eng.set_dropout(True, seed)            # resets the pass counter: identical masks on every call
    eng.forward(0, use_fast=use_fast, train=True)
    l = float(eng.loss(0)[0, 0])
    eng.backward(0, use_fast=use_fast, scale=1.0, need_encoder=True)
    return l, {n: eng.export(n, 2, 0).astype(np.float64) for n in names}


def test_dropout_masks_are_replayed_and_gradients_consistent(emu_lib):
    """Train-mode dropout (SubLayers.py:54,90; modules.py:223,235; Layers.py:133-134): counter-based masks, replayed by
    backward.  No bit-parity with torch's RNG is possible, so: determinism per seed, sensitivity to the seed, keep-rate, and
    a central finite-difference check of the analytic gradient with the masks frozen."""
    dims = tiny_dims()
    eng = _engine(dims, emu_lib, tasks=1)
    b0 = synth.make_batch(3, 3, speaker=2, **_kw(dims))
    eng.set_batches(0, [b0])
    names = ["decoder.layer_stack.1.pos_ffn.w_2.weight", "variance_adaptor.pitch_predictor.conv_layer.conv1d_2.conv.weight",
             "postnet.convolutions.1.0.conv.weight", "encoder.layer_stack.0.slf_attn.fc.weight", "mel_linear.weight"]
    l1, g1 = _loss_and_grads(eng, 7, names)
    l2, g2 = _loss_and_grads(eng, 7, names)
    l3, _ = _loss_and_grads(eng, 8, names)
    assert l1 == l2 and all(np.array_equal(g1[n], g2[n]) for n in names)
    assert l1 != l3
    eng.set_dropout(False)
    eng.forward(0, train=True)
    l0 = float(eng.loss(0)[0, 0])
    assert abs(l1 - l0) > 1e-3  # dropout really changes the forward
    # keep-rate of the last PostNet layer's mask: zeros of mel_post - mel where the un-dropped value is non-zero
    eng.set_dropout(True, 7)
    eng.forward(0, train=True)
    o = eng.outputs(0, 0)
    resid = o["mel_post"] - o["mel"]
    frac_zero = float((resid == 0).mean())
    assert 0.4 < frac_zero < 0.6
    # finite differences along a random direction, masks frozen by re-seeding
    base = {n: eng.export(n) for n in names}
    g = np.random.RandomState(0)
    dirs = {n: g.standard_normal(base[n].shape).astype(np.float32) for n in names}
    eps = 2e-4  # small enough to stay on one side of the ReLU / L1 kinks; fp32 loss noise then is ~0.1 in the quotient
    for n in names:
        vals = []
        for sgn in (+1, -1):
            eng.load_params({n: base[n] + sgn * eps * dirs[n]}, strict=False)
            vals.append(_loss_and_grads(eng, 7, names)[0])
        eng.load_params({n: base[n]}, strict=False)
        fd = (vals[0] - vals[1]) / (2 * eps)
        an = float((g1[n] * dirs[n]).sum())
        assert abs(fd - an) <= 0.05 * abs(an) + 0.3, (n, fd, an)
    eng.close()


def test_second_order_with_dropout_replays_inner_step_masks(emu_lib):
    """HVP with dropout on.  fp32 finite differences of gradients are too noisy for a second derivative, so the check is
    exact algebra instead: with frozen masks the loss is a smooth function, its Hessian block over the adapted weights is
    symmetric, hence <u, H v> == <v, H u> for two unrelated directions — which only holds if the tangent forward and the
    tangent backward regenerate exactly the masks of the primal pass."""
    dims = tiny_dims()
    eng = _engine(dims, emu_lib, tasks=1)
    b0 = synth.make_batch(3, 3, speaker=2, **_kw(dims))
    eng.set_batches(0, [b0])
    adapted = [n for n in eng.params if eng.params[n][2]]
    base = {n: eng.export(n) for n in adapted}
    g = np.random.RandomState(3)

    def direction_and_hv(perturb):
        # direction = gradient at (theta + perturb) left in the gradient buffer; H evaluated at theta
        eng.load_params({n: base[n] + perturb[n] for n in adapted}, strict=False)
        eng.adapt(0, 0.0, reset=True)
        _, v = _loss_and_grads(eng, 11, adapted, use_fast=True)
        eng.load_params(base, strict=False)
        eng.adapt(0, 0.0, reset=True)
        eng.set_dropout(True, 11)  # the HVP re-runs the forward: counter reset => same masks as every other pass here
        eng.hvp_support()
        return v, {n: eng.export(n, 6, 0).astype(np.float64) for n in adapted}

    zero = {n: np.zeros_like(base[n]) for n in adapted}
    bump = {n: (0.02 * g.standard_normal(base[n].shape) * np.abs(base[n]).mean()).astype(np.float32) for n in adapted}
    u, Hu = direction_and_hv(zero)
    v, Hv = direction_and_hv(bump)
    uHv = sum(float((u[n] * Hv[n]).sum()) for n in adapted)
    vHu = sum(float((v[n] * Hu[n]).sum()) for n in adapted)
    assert abs(uHv - vHu) <= 2e-3 * max(abs(uHv), abs(vHu)), (uHv, vHu)
    # and the masks matter: without replay (different seed for the HVP) symmetry is lost
    eng.load_params(base, strict=False)
    eng.adapt(0, 0.0, reset=True)
    _loss_and_grads(eng, 11, adapted, use_fast=True)
    eng.set_dropout(True, 12)
    eng.hvp_support()
    Hu_wrong = {n: eng.export(n, 6, 0).astype(np.float64) for n in adapted}
    assert abs(sum(float((v[n] * Hu_wrong[n]).sum()) for n in adapted) - uHv) > 1e-2 * abs(uHv)
    eng.close()


def test_adapted_encoder_moves_in_the_inner_loop(emu_lib):
    """config/algorithm/dev.yaml lists `encoder` in adapt.modules: the inner-loop backward must reach the encoder so its
    fast weights (incl. the phoneme table) move, as learn2learn's adapt does for every cloned module (utils.py:17-77)."""
    dims = tiny_dims()
    mods = ["encoder", "decoder"]
    eng = _engine(dims, emu_lib, tasks=1, mods=mods)
    sup = synth.make_batch(30, 2, speaker=3, **_kw(dims)); qry = synth.make_batch(31, 2, speaker=3, **_kw(dims))
    eng.set_batches(0, [sup]); eng.set_batches(1, [qry], spk_from=[sup], average_spk=True)
    q, s = eng.meta_grad(2, 0.05, 1.0)
    p = torch_params(dims, requires_grad=True)
    ql, sl, fast, _ = O.maml_task(p, torch_buffers(dims), O.to_torch_batch(sup), O.to_torch_batch(qry), steps=2, lr=0.05,
                                  second_order=False, modules=mods, n_head=heads(dims), max_seq_len=dims.max_seq_len)
    np.testing.assert_allclose(q[0], [float(x) for x in ql], rtol=1e-4)
    np.testing.assert_allclose(s[:, 0, 0], [float(l[0]) for l in sl], rtol=1e-4)
    moved = 0.0
    for n in O.adapted_names(p, mods):
        ref = fast[n].detach().numpy()
        got = eng.export(n, 3, 0)
        assert np.abs(got - ref).max() <= 2e-5 * max(np.abs(ref).max(), 1e-3), n
        if n.startswith("encoder."):
            moved += float(np.abs(got - eng.export(n, 0)).sum())
    assert moved > 1e-3  # the encoder's fast weights really left theta
    for n in ("encoder.layer_stack.0.pos_ffn.w_1.weight", "encoder.src_word_emb.weight", "mel_linear.weight"):
        g = torch.autograd.grad(ql[0], p[n], retain_graph=True)[0].numpy()
        assert np.abs(eng.export(n, 1) - g).max() <= 2e-3 * np.abs(g).max() + 2e-7, n
    # the same through mtts_adapt (few-shot test loop)
    eng.adapt(2, 0.05, reset=True)
    n = "encoder.layer_stack.0.slf_attn.fc.weight"
    assert np.abs(eng.export(n, 3, 0) - fast[n].detach().numpy()).max() <= 2e-5 * np.abs(fast[n].detach().numpy()).max()
    eng.meta_grad(2, 0.05, 1.0, second_order=True)  # supported since round 3 (checked against the oracle in test_second_order_maml_with_adapted_encoder)
    eng.close()


def test_out_of_range_token_ids_are_rejected(emu_lib):
    dims = tiny_dims()
    eng = _engine(dims, emu_lib, tasks=1)
    b = list(synth.make_batch(3, 2, speaker=2, **_kw(dims)))
    for bad in (dims.vocab, -1):
        t = b[3].copy(); t[0, 0] = bad
        with pytest.raises(Exception, match="token"):
            eng.set_batches(0, [tuple(b[:3]) + (t,) + tuple(b[4:])])
    t = b[3].copy(); t[1, int(b[4][1]):] = 10 ** 6   # padding positions are never read
    eng.set_batches(0, [tuple(b[:3]) + (t,) + tuple(b[4:])])
    eng.close()


def test_two_handles_on_two_host_threads_do_not_interfere(emu_lib):
    """include/mtts.h conventions: different handles share no mutable state (launch-batching queue, profiler, split-K
    workspace, numerics mode are per handle), so two host threads may drive two handles at once — results must equal the
    single-threaded ones bit for bit."""
    import threading
    dims = tiny_dims()
    jobs = [(3, 2, 0.01), (4, 5, 0.02)]

    def run(seed, spk, lr, out, key):
        eng = _engine(dims, emu_lib, tasks=1)
        sup = synth.make_batch(seed, 3, speaker=spk, **_kw(dims)); qry = synth.make_batch(seed + 10, 2, speaker=spk, **_kw(dims))
        for _ in range(2):
            eng.set_batches(0, [sup]); eng.set_batches(1, [qry], spk_from=[sup], average_spk=True)
            q, s = eng.meta_grad(2, lr, 1.0)
        out[key] = (q.copy(), s.copy(), eng.export("decoder.layer_stack.1.pos_ffn.w_1.weight", 1), eng.export("mel_linear.weight", 1))
        eng.close()

    serial, threaded = {}, {}
    for i, (seed, spk, lr) in enumerate(jobs):
        run(seed, spk, lr, serial, i)
    ts = [threading.Thread(target=run, args=(seed, spk, lr, threaded, i)) for i, (seed, spk, lr) in enumerate(jobs)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    for i in range(len(jobs)):
        for a, b in zip(serial[i], threaded[i]):
            np.testing.assert_array_equal(a, b)


def test_imaml_hypergradient_matches_oracle(emu_lib):
    _check_imaml(emu_lib, MODS)


def test_imaml_hypergradient_with_adapted_encoder(emu_lib):
    """config/algorithm/dev.yaml:22-33: `type: imaml` with the encoder among the adapted modules."""
    _check_imaml(emu_lib, ENC_MODS)


def _check_imaml(emu_lib, mods):
    """iMAML (imaml.py:41-139, utils.py:120-189): proximal first-order inner loop on support mini-batches, then K conjugate-gradient
    iterations on a (H + reg I) with a fresh mini-batch per Hessian-vector product (`stochastic: true`), per-task clip, and the
    hypergradient a * reg * v on the adapted parameters only — two tasks in one grouped pass against the torch restatement."""
    from meta_tts_amd import data as D
    dims = tiny_dims()
    MODS = mods
    eng = _engine(dims, emu_lib, mods=mods)
    lr, reg, K = 0.02, 1.0, 3
    tasks = [(synth.make_batch(50 + 2 * j, 3, speaker=2 + j, **_kw(dims)), synth.make_batch(51 + 2 * j, 2, speaker=2 + j, **_kw(dims)))
             for j in range(2)]
    sub = lambda b, idx: D.split_reprocess(b, idx)     # Task.next_batch (systems/utils.py:80-117): a re-cropped sub-batch
    inner_idx = [[0, 1], [2, 0]]                       # 2 inner steps on 2-utterance mini-batches
    cg_idx = [[1, 2], [0, 2], [0, 1]]                  # K mini-batches for the Hessian-vector products
    eng.set_inner_prox(reg)
    first = True
    for idx in inner_idx:
        eng.set_batches(0, [sub(t[0], idx) for t in tasks])
        eng.adapt(1, lr, reset=first)
        first = False
    eng.set_batches(0, [t[0] for t in tasks])          # the speaker ids of the query pass come from the whole support set
    eng.set_batches(1, [t[1] for t in tasks], spk_from=[t[0] for t in tasks], average_spk=True)
    q = eng.imaml_begin()
    for idx in cg_idx:
        eng.set_batches(0, [sub(t[0], idx) for t in tasks])
        eng.imaml_cg_step(lr, reg)
    norms = eng.imaml_finish(lr, reg, grad_scale=0.5, max_norm=0.0)
    tot, per_task = None, []
    for j, (sup, qry) in enumerate(tasks):
        p = torch_params(dims, requires_grad=True)
        tsup = O.to_torch_batch(sup)
        ql, fast, hyper, v = O.imaml_task(p, torch_buffers(dims), [O.to_torch_batch(sub(sup, i)) for i in inner_idx],
                                          [O.to_torch_batch(sub(sup, i)) for i in cg_idx], O.to_torch_batch(qry), tsup[2], lr=lr, reg_param=reg,
                                          K=K, modules=MODS, n_head=heads(dims), max_seq_len=dims.max_seq_len)
        np.testing.assert_allclose(q[j], [float(x) for x in ql], rtol=1e-4)
        for n in ("mel_linear.weight", "decoder.layer_stack.0.pos_ffn.w_1.weight"):
            ref = fast[n].detach().numpy()
            assert np.abs(eng.export(n, 3, j) - ref).max() <= 2e-5 * np.abs(ref).max(), n     # proximal inner loop
        ref_norm = float(np.sqrt(sum(float((h.double() ** 2).sum()) for h in hyper.values())))
        assert abs(norms[j] - ref_norm) <= 5e-3 * ref_norm
        g = {n: 0.5 * h.numpy() for n, h in hyper.items()}
        per_task.append({n: h.numpy() for n, h in hyper.items()})
        tot = g if tot is None else {n: tot[n] + g[n] for n in g}
    scale = max(np.abs(x).max() for x in tot.values())
    for n in eng.params:
        got = eng.export(n, 1)
        if n in tot:
            assert np.abs(got - tot[n]).max() <= 5e-3 * np.abs(tot[n]).max() + 1e-6 * scale, n
        else:
            assert not got.any(), n           # non-adapted parameters receive no hypergradient (utils.py:147-150)
    # per-task clipping before the reduction (imaml.py:125-131)
    clip = 0.5 * float(norms.min())
    eng.imaml_finish(lr, reg, grad_scale=1.0, max_norm=clip)
    n = "mel_linear.weight"
    want = sum(min(1.0, clip / (float(norms[j]) + 1e-6)) * per_task[j][n] for j in range(2))
    got = eng.export(n, 1)
    assert np.abs(got - want).max() <= 5e-3 * np.abs(want).max()
    assert min(1.0, clip / float(norms.max())) < 0.6   # both tasks really were clipped
    # the proximal term must be switched off again for plain MAML, and second order refuses it
    with pytest.raises(Exception):
        eng.meta_grad(1, lr, 1.0, second_order=True)
    eng.set_inner_prox(0.0)
    eng.close()


@pytest.mark.parametrize("pl,el", [("frame_level", "frame_level"), ("phoneme_level", "frame_level"), ("frame_level", "phoneme_level")])
def test_frame_level_pitch_energy_matches_oracle(emu_lib, pl, el):
    """preprocess `pitch.feature / energy.feature: frame_level` (modules.py:139-148, loss.py:54-63): predictor, bucketised embedding
    and loss of that feature on the frame rectangle after the length regulator — forward, 6 losses, every parameter gradient,
    first-order MAML and free-running synthesis with controls, two ragged tasks, all three level combinations."""
    dims = tiny_dims(pitch_level=pl, energy_level=el)
    eng = _engine(dims, emu_lib)
    kw = dict(pitch_level=pl, energy_level=el, **_kw(dims))
    b0 = synth.make_batch(3, 3, speaker=2, **kw)
    b1 = synth.make_batch(4, 2, speaker=5, **kw)
    assert b0[9].shape == (3, b0[8] if pl == "frame_level" else b0[5])
    eng.set_batches(0, [b0, b1])
    eng.forward(0, use_fast=False, train=True)
    dev_loss = eng.loss(0)
    eng.backward(0, use_fast=False, scale=1.0, need_encoder=True)
    okw = dict(n_head=heads(dims), max_seq_len=dims.max_seq_len, pitch_level=pl, energy_level=el)
    for ti, b in enumerate([b0, b1]):
        p = torch_params(dims, requires_grad=True)
        tb = O.to_torch_batch(b)
        o = O.fs2_forward(p, torch_buffers(dims), *tb[2:], training=True, **okw)
        lo = O.fs2_loss(tb, o, pl, el)
        out = eng.outputs(0, ti)
        for k, ref in (("mel", o[0]), ("mel_post", o[1]), ("p", o[2]), ("e", o[3]), ("logd", o[4])):
            assert out[k].shape == tuple(ref.shape), (k, out[k].shape, ref.shape)
            assert np.abs(out[k] - ref.detach().numpy()).max() < 5e-5, (ti, k)
        np.testing.assert_allclose(dev_loss[ti], [float(x) for x in lo], rtol=2e-5)
        names = list(eng.params)
        gs = torch.autograd.grad(lo[0], [p[n] for n in names], allow_unused=True)
        for n, g in zip(names, gs):
            ref = g.numpy() if g is not None else np.zeros(eng.params[n][0], np.float32)
            assert np.abs(eng.export(n, 2, ti) - ref).max() <= 1e-3 * np.abs(ref).max() + 2e-7, (ti, n)
    # first-order MAML on top of it
    q0 = synth.make_batch(13, 2, speaker=2, **kw); q1 = synth.make_batch(14, 2, speaker=5, **kw)
    eng.set_batches(1, [q0, q1], spk_from=[b0, b1], average_spk=True)
    q, s_ = eng.meta_grad(2, 0.01, 0.5)
    tot = {}
    check = ["variance_adaptor.pitch_predictor.conv_layer.conv1d_1.conv.weight", "variance_adaptor.energy_embedding.weight", "mel_linear.weight",
             "encoder.layer_stack.0.pos_ffn.w_1.weight"]
    for j, (sup, qry) in enumerate([(b0, q0), (b1, q1)]):
        p = torch_params(dims, requires_grad=True)
        names = O.adapted_names(p, MODS)
        fast = {k: p[k] for k in names}
        ts, tq = O.to_torch_batch(sup), O.to_torch_batch(qry)
        for _ in range(2):
            cur = dict(p); cur.update(fast)
            l = O.fs2_loss(ts, O.fs2_forward(cur, torch_buffers(dims), *ts[2:], training=True, **okw), pl, el)
            g = torch.autograd.grad(l[0], [fast[k] for k in names])
            fast = {k: fast[k] - 0.01 * gi for k, gi in zip(names, g)}
        cur = dict(p); cur.update(fast)
        ql = O.fs2_loss(tq, O.fs2_forward(cur, torch_buffers(dims), ts[2], *tq[3:], training=True, average_spk_emb=True, **okw), pl, el)
        np.testing.assert_allclose(q[j], [float(x) for x in ql], rtol=1e-4)
        for n, x in zip(check, torch.autograd.grad(ql[0], [p[n] for n in check])):
            tot[n] = tot.get(n, 0) + 0.5 * x.numpy()
    for n in check:
        assert np.abs(eng.export(n, 1) - tot[n]).max() <= 2e-3 * np.abs(tot[n]).max() + 2e-7, n
    # second order runs on frame-level features too (its numbers: test_frame_level_hessian_vector_product)
    q2, _ = eng.meta_grad(1, 0.01, 0.5, second_order=True)
    assert np.isfinite(q2).all()
    # free-running with controls
    params = synth.make_params(dims, 0)
    params["variance_adaptor.duration_predictor.linear_layer.bias"][:] = 1.2
    eng.close()
    eng = _engine(dims, emu_lib)       # fresh BatchNorm running statistics for the eval-mode pass
    eng.load_params(params)
    pt = {k: torch.from_numpy(v.copy()) for k, v in params.items()}
    for train in (False, True):
        eng.set_batches(0, [b0[:6], b1[:6]])
        eng.synthesize(0, train=train, p_control=1.1, e_control=0.9)
        for ti, b in enumerate([b0, b1]):
            tb = O.to_torch_batch(b)
            with torch.no_grad():
                o = O.fs2_forward(pt, torch_buffers(dims), *tb[2:6], p_control=1.1, e_control=0.9, training=train, **okw)
            out = eng.outputs(0, ti)
            np.testing.assert_array_equal(out["d_rounded"], o[5].numpy())
            assert out["mel_post"].shape == tuple(o[1].shape) and np.abs(out["mel_post"] - o[1].numpy()).max() < 5e-5
            np.testing.assert_allclose(out["p"] * 1.1, o[2].numpy(), atol=5e-5)
            np.testing.assert_allclose(out["e"] * 0.9, o[3].numpy(), atol=5e-5)
    eng.close()


def test_external_speaker_embeddings_match_a_table_of_the_same_rows(emu_lib):
    """`speaker_emb: dvec` (speaker_encoder.py:71-76 -> fastspeech2.py:65-68,91-94): the batch carries one (d_model) embedding per
    utterance instead of a speaker id.  Against the oracle with those embeddings as the rows of the table and ids 0..B-1 (the same
    arithmetic): outputs, losses and every non-speaker gradient agree; the (placeholder) table receives no gradient."""
    dims = tiny_dims()
    eng = _engine(dims, emu_lib, tasks=2)
    g = np.random.RandomState(5)
    bs = [synth.make_batch(3, 3, speaker=0, **_kw(dims)), synth.make_batch(4, 2, speaker=0, **_kw(dims))]
    embs = [g.standard_normal((len(b[4]), dims.d_model)).astype(np.float32) for b in bs]
    ext = [tuple(b[:2]) + (e,) + tuple(b[3:]) for b, e in zip(bs, embs)]
    eng.set_batches(0, ext)
    eng.forward(0, use_fast=False, train=True)
    dev_loss = eng.loss(0)
    eng.backward(0, use_fast=False, scale=1.0, need_encoder=True)
    for ti, (b, e) in enumerate(zip(bs, embs)):
        p = torch_params(dims, requires_grad=True)
        p["speaker_emb.model.weight"] = torch.from_numpy(e.copy()).requires_grad_(True)
        tb = list(O.to_torch_batch(b))
        tb[2] = torch.arange(e.shape[0])
        o = O.fs2_forward(p, torch_buffers(dims), *tb[2:], n_head=heads(dims), max_seq_len=dims.max_seq_len, training=True)
        lo = O.fs2_loss(tuple(tb), o)
        out = eng.outputs(0, ti)
        for k, ref in (("mel", o[0]), ("mel_post", o[1]), ("p", o[2]), ("e", o[3]), ("logd", o[4])):
            assert np.abs(out[k] - ref.detach().numpy()).max() < 5e-5, (ti, k)
        np.testing.assert_allclose(dev_loss[ti], [float(x) for x in lo], rtol=2e-5)
        names = [n for n in eng.params if n != "speaker_emb.model.weight"]
        gs = torch.autograd.grad(lo[0], [p[n] for n in names], allow_unused=True)
        for n, gr in zip(names, gs):
            ref = gr.numpy() if gr is not None else np.zeros(eng.params[n][0], np.float32)
            assert np.abs(eng.export(n, 2, ti) - ref).max() <= 1e-3 * np.abs(ref).max() + 2e-7, (ti, n)
        assert not eng.export("speaker_emb.model.weight", 2, ti).any()
    # Hessian-vector products with external embeddings (round 3): the embeddings carry no tangent, the table receives none
    eng.adapt(0, 0.0, reset=True)
    eng.forward(0, use_fast=True, train=True)
    eng.backward(0, use_fast=True, scale=1.0, need_encoder=True)
    eng.hvp_support()
    for ti, (b, e) in enumerate(zip(bs, embs)):
        p = torch_params(dims, requires_grad=True)
        p["speaker_emb.model.weight"] = torch.from_numpy(e.copy())
        tb = list(O.to_torch_batch(b))
        tb[2] = torch.arange(e.shape[0])
        lo = O.fs2_loss(tuple(tb), O.fs2_forward(p, torch_buffers(dims), *tb[2:], n_head=heads(dims), max_seq_len=dims.max_seq_len, training=True))
        an = [n for n in O.adapted_names(p, MODS) if n != "speaker_emb.model.weight"]
        gsup = torch.autograd.grad(lo[0], [p[n] for n in an], create_graph=True)
        dot = sum((gi * gi.detach()).sum() for gi in gsup)
        names = [n for n in eng.params if n != "speaker_emb.model.weight"]
        hv = torch.autograd.grad(dot, [p[n] for n in names], allow_unused=True)
        scale = max(float(h.abs().max()) for h in hv if h is not None)
        for n, h in zip(names, hv):
            ref = h.numpy() if h is not None else np.zeros(eng.params[n][0], np.float32)
            assert np.abs(eng.export(n, 6, ti) - ref).max() <= 2e-3 * np.abs(ref).max() + 1e-7 * scale, (ti, n)
    # mixing embedded and id batches in one call is rejected loudly
    with pytest.raises(Exception, match="every task or none"):
        eng.set_batches(0, [ext[0], bs[1]])
    eng.close()


@pytest.mark.parametrize("pl,el,mods", [("frame_level", "frame_level", MODS), ("phoneme_level", "frame_level", MODS),
                                        ("frame_level", "phoneme_level", ENC_MODS)],
                         ids=["both_frame", "energy_frame", "pitch_frame_adapted_encoder"])
def test_frame_level_hessian_vector_product(emu_lib, pl, el, mods):
    """Second order with frame-level pitch / energy (modules.py:139-148 under create_graph=True): the forward-over-reverse HVP with the
    variance adaptor's second half on the frame rectangle, against torch's double backward, every tensor, two ragged tasks — incl. the
    combination with an adapted encoder (primal input gradients through the rectangle)."""
    dims = tiny_dims(pitch_level=pl, energy_level=el)
    eng = _engine(dims, emu_lib, mods=mods)
    kw = dict(pitch_level=pl, energy_level=el, **_kw(dims))
    b0 = synth.make_batch(3, 3, speaker=2, **kw); b1 = synth.make_batch(4, 2, speaker=5, **kw)
    eng.set_batches(0, [b0, b1])
    eng.adapt(0, 0.0, reset=True)  # fast weights := theta
    eng.forward(0, use_fast=True, train=True)
    eng.backward(0, use_fast=True, scale=1.0, need_encoder=True)
    eng.hvp_support()
    okw = dict(n_head=heads(dims), max_seq_len=dims.max_seq_len, pitch_level=pl, energy_level=el)
    for ti, b in enumerate([b0, b1]):
        p = torch_params(dims, requires_grad=True)
        tb = O.to_torch_batch(b)
        lo = O.fs2_loss(tb, O.fs2_forward(p, torch_buffers(dims), *tb[2:], training=True, **okw), pl, el)
        an = O.adapted_names(p, mods)
        g = torch.autograd.grad(lo[0], [p[n] for n in an], create_graph=True)
        dot = sum((gi * gi.detach()).sum() for gi in g)
        names = list(eng.params)
        hv = torch.autograd.grad(dot, [p[n] for n in names], allow_unused=True)
        scale = max(float(h.abs().max()) for h in hv if h is not None)
        for n, h in zip(names, hv):
            ref = h.numpy() if h is not None else np.zeros(eng.params[n][0], np.float32)
            got = eng.export(n, 6, ti)
            assert np.abs(got - ref).max() <= 2e-3 * np.abs(ref).max() + 1e-7 * scale, (ti, n)
    # and the whole second-order meta-gradient runs on it
    q0 = synth.make_batch(13, 2, speaker=2, **kw); q1 = synth.make_batch(14, 2, speaker=5, **kw)
    eng.set_batches(1, [q0, q1], spk_from=[b0, b1], average_spk=True)
    q, _ = eng.meta_grad(2, 0.01, 0.5, second_order=True)
    assert np.isfinite(q).all()
    eng.close()
