"""The fp64 arbiter of gradient parity (oracle/arbiter.py) on the SIMT emulator (CPU): it passes the engine on a dropout-on meta-gradient,
prices L1 sign flips and ReLU flips exactly (a party built from the fp64 gradient with a known flip is explained to roundoff), and refuses a
dense 2 % error — the escape the old kink rule (relative L2 <= 3e-2 for any tensor of a task with a kink) left open."""
import numpy as np
import pytest
import torch

import __graft_entry__ as ge
from oracle_util import O, heads, synth, tiny_dims, torch_buffers, torch_params
from oracle import arbiter as A
from oracle.dropout_masks import DropoutMasks, plan_seed
from meta_tts_amd.engine import Engine

MODS = ["speaker_emb", "variance_adaptor", "decoder", "mel_linear", "postnet"]
PROBS = dict(enc=0.2, dec=0.2, vp=0.5, postnet=0.5)
STEPS, LR, SEED = 3, 1e-4, 5


@pytest.fixture(scope="module")
def setup():
    lib = ge.build_emulator()
    dims = tiny_dims()
    kw = dict(n_mel=dims.n_mel, vocab=dims.vocab, s_range=(5, 13), d_range=(1, 6), first_len=12)
    sup, qry = synth.make_batch(31, 3, speaker=2, **kw), synth.make_batch(32, 2, speaker=2, **kw)
    # put three query mel targets within 2e-6 of where the prediction lands: L1 kinks inside fp32 noise, as at full size
    p = torch_params(dims, requires_grad=True)
    dms = [DropoutMasks(plan_seed(SEED, k + 1), 0, PROBS) for k in range(STEPS + 1)]
    ql, _, _, preds = O.maml_task(p, torch_buffers(dims), O.to_torch_batch(sup), O.to_torch_batch(qry), steps=STEPS, lr=LR, second_order=False,
                                  modules=MODS, n_head=heads(dims), max_seq_len=dims.max_seq_len, dropout=dms)
    qry = list(qry)
    qry[6] = qry[6].copy()
    planted = ((0, 1, 3), (1, 2, 7), (0, 4, 0))
    for (b, t, c), off in zip(planted, (2e-6, -1e-6, 5e-7)):
        qry[6][b, t, c] = np.float32(float(preds[1][b, t, c]) + off)
    qry = tuple(qry)
    eng = Engine(dims, adapt_modules=MODS, max_tasks=1, max_B=3, max_S=16, max_T=96, lib_path=lib)
    np_params = synth.make_params(dims, 0)
    eng.load_params(np_params)
    eng.set_batches(0, [sup])
    eng.set_batches(1, [qry], spk_from=[sup], average_spk=True)
    eng.set_dropout(True, SEED)
    eng.meta_grad(STEPS, LR, 1.0)
    names = [n for n in eng.params if not n.startswith("encoder.")][:40:3] + ["encoder.layer_stack.0.pos_ffn.w_1.weight", "mel_linear.weight",
                                                                              "postnet.convolutions.0.0.conv.weight", "postnet.convolutions.2.1.weight"]
    out = eng.outputs(1, 0)
    engine = {"grads": {n: eng.export(n, 2, 0) for n in names}, "mel": out["mel"], "mel_post": out["mel_post"]}
    eng.close()
    # the fp32 oracle on the patched task
    p = torch_params(dims, requires_grad=True)
    dms = [DropoutMasks(plan_seed(SEED, k + 1), 0, PROBS) for k in range(STEPS + 1)]
    ql, _, _, preds = O.maml_task(p, torch_buffers(dims), O.to_torch_batch(sup), O.to_torch_batch(qry), steps=STEPS, lr=LR, second_order=False,
                                  modules=MODS, n_head=heads(dims), max_seq_len=dims.max_seq_len, dropout=dms)
    gs = torch.autograd.grad(ql[0], [p[n] for n in names], allow_unused=True)
    oracle32 = {"grads": {n: (g.numpy() if g is not None else np.zeros(tuple(p[n].shape), np.float32)) for n, g in zip(names, gs)},
                "mel": preds[0].detach().numpy(), "mel_post": preds[1].detach().numpy()}

    def run(parties, **kw_):
        dms = [DropoutMasks(plan_seed(SEED, k + 1), 0, PROBS) for k in range(STEPS + 1)]
        return A.arbitrate_task(np_params, synth.make_buffers(dims), sup, qry, modules=MODS, n_head=heads(dims), max_seq_len=dims.max_seq_len,
                                steps=STEPS, lr=LR, masks=dms, names=names, parties=parties, **kw_)
    return dict(run=run, engine=engine, oracle32=oracle32, names=names, dims=dims, sup=sup, qry=qry, np_params=np_params, planted=planted)


def test_engine_outputs_after_meta_grad_are_the_query_forward(setup):
    """The arbiter takes each party's L1 signs from its own query-pass mel / mel_post: the engine's must still be there after the backward."""
    e, o = setup["engine"], setup["oracle32"]
    T = e["mel"].shape[1]
    assert np.abs(e["mel_post"] - o["mel_post"][:, :T]).max() < 1e-4
    assert np.abs(e["mel"] - o["mel"][:, :T]).max() < 1e-4


def test_arbiter_passes_the_engine_and_reports_both_errors(setup):
    rep = setup["run"]({"engine": setup["engine"], "oracle32": setup["oracle32"]})
    assert rep["pass"], rep
    assert rep["l1_ambiguous_elements"] >= 3
    for n, row in rep["tensors"].items():
        assert set(row["engine"]) >= {"raw", "l1"} and set(row["oracle32"]) >= {"raw", "l1"}
        assert row["engine"]["l1"] < 2e-3, (n, row)          # fp32 roundoff against fp64 once the signs are the party's own


def test_a_flipped_l1_sign_is_priced_exactly(setup):
    """A party that is EXACT (float64) except that its forward output sits on the other side of the target at the three planted elements — its
    gradient computed independently here by moving those targets across the prediction: raw error = those units' whole contribution, `l1`
    error at float64 roundoff."""
    dims, sup, qry, np_params = setup["dims"], setup["sup"], setup["qry"], setup["np_params"]
    names = setup["names"]

    def f64_grad(qry_):
        p = A.f64_params(np_params)
        buf = {k: torch.from_numpy(v.copy()).double() if v.dtype.kind == "f" else torch.from_numpy(v.copy()) for k, v in synth.make_buffers(dims).items()}
        dms = [DropoutMasks(plan_seed(SEED, k + 1), 0, PROBS) for k in range(STEPS + 1)]
        ql, _, _, preds = O.maml_task(p, buf, A.f64_batch(sup), A.f64_batch(qry_), steps=STEPS, lr=LR, second_order=False, modules=MODS,
                                      n_head=heads(dims), max_seq_len=dims.max_seq_len, dropout=dms)
        gs = torch.autograd.grad(ql[0], [p[n] for n in names], allow_unused=True)
        return {n: (g.numpy() if g is not None else np.zeros(tuple(p[n].shape))) for n, g in zip(names, gs)}, preds
    g_true, preds = f64_grad(qry)
    mel_post = preds[1].detach().numpy().copy()
    moved = [q.copy() if isinstance(q, np.ndarray) else q for q in qry]
    mirrored = mel_post.copy()
    for (b, t, c) in setup["planted"]:
        r = mel_post[b, t, c] - float(qry[6][b, t, c])
        moved[6][b, t, c] = np.float32(mel_post[b, t, c] + 10.0 * r + np.sign(r) * 1e-3)     # the target crosses the prediction: sign flips, nothing else moves
        mirrored[b, t, c] = float(qry[6][b, t, c]) - r
    g_flip, _ = f64_grad(tuple(moved))
    # (the PostNet's last dropout zeroes half of its output: there mel_post == mel and the mel term crosses its target as well)
    melv = preds[0].detach().numpy().copy()
    flips = len(setup["planted"])
    for (b, t, c) in setup["planted"]:
        if np.sign(melv[b, t, c] - float(qry[6][b, t, c])) != np.sign(melv[b, t, c] - float(moved[6][b, t, c])):
            melv[b, t, c] = 2.0 * float(qry[6][b, t, c]) - melv[b, t, c]
            flips += 1
    party = {"grads": g_flip, "mel": melv, "mel_post": mirrored}
    rep = setup["run"]({"engine": party}, explain=False)
    assert rep["parties"]["engine"]["l1_flips"] == flips
    post = [n for n in names if n.startswith("postnet")]
    assert max(rep["tensors"][n]["engine"]["raw"] for n in post) > 1e-3          # the flips are visible ...
    for n in names:
        assert rep["tensors"][n]["engine"]["l1"] < 1e-9, (n, rep["tensors"][n])   # ... and priced to float64 roundoff
    exact = setup["run"]({"engine": {"grads": g_true, "mel": preds[0].detach().numpy(), "mel_post": mel_post}}, explain=False)
    assert exact["parties"]["engine"]["l1_flips"] == 0 and all(exact["tensors"][n]["engine"]["raw"] < 1e-9 for n in names)


def test_arbiter_refuses_a_dense_two_percent_error(setup):
    bad = {"grads": {n: g * (1.02 if n.startswith("decoder") else 1.0) for n, g in setup["engine"]["grads"].items()},
           "mel": setup["engine"]["mel"], "mel_post": setup["engine"]["mel_post"]}
    rep = setup["run"]({"engine": bad, "oracle32": setup["oracle32"]})
    assert not rep["pass"]
    failed = [n for n, row in rep["tensors"].items() if not row["ok"]]
    assert failed and all(n.startswith("decoder") for n in failed), failed


def test_flipped_relu_units_are_identified_and_priced(setup, monkeypatch):
    """A party that is EXACT (float64) except that its BACKWARD treats a few ReLU units with the smallest pre-activations of one decoder FFN
    the other way (forward values untouched — what two fp32 implementations on different sides of zero look like): the arbiter finds exactly
    those units among the ambiguous ones and the residual drops from the units' whole contribution to float64 roundoff."""
    dims, sup, qry, np_params, names = setup["dims"], setup["sup"], setup["qry"], setup["np_params"], setup["names"]
    per_pass = dims.enc_layers + dims.dec_layers + 6
    target_call = STEPS * per_pass + dims.enc_layers + 6        # the query pass's first decoder FFN (encoder, 3 predictors x 2, decoder)
    state = {"n": 0, "flip": None, "x": None}

    class FlipRelu(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, flip):
            ctx.save_for_backward((x > 0) ^ flip)
            return torch.relu(x)

        @staticmethod
        def backward(ctx, g):
            (m,) = ctx.saved_tensors
            return g * m, None

    def relu(x):
        k = state["n"]
        state["n"] += 1
        if k == target_call:
            state["x"] = x.detach()
            if state["flip"] is not None:
                return FlipRelu.apply(x, state["flip"])
        return torch.relu(x)

    def f64_grad():
        state["n"] = 0
        p = A.f64_params(np_params)
        buf = {k: torch.from_numpy(v.copy()).double() if v.dtype.kind == "f" else torch.from_numpy(v.copy()) for k, v in synth.make_buffers(dims).items()}
        dms = [DropoutMasks(plan_seed(SEED, k + 1), 0, PROBS) for k in range(STEPS + 1)]
        ql, _, _, preds = O.maml_task(p, buf, A.f64_batch(sup), A.f64_batch(qry), steps=STEPS, lr=LR, second_order=False, modules=MODS,
                                      n_head=heads(dims), max_seq_len=dims.max_seq_len, dropout=dms)
        gs = torch.autograd.grad(ql[0], [p[n] for n in names], allow_unused=True)
        return {n: (g.numpy() if g is not None else np.zeros(tuple(p[n].shape))) for n, g in zip(names, gs)}, preds
    monkeypatch.setattr(O, "_relu", relu)
    g_true, preds = f64_grad()
    x = state["x"]
    # the 4 smallest |pre-activations| at VALID frames (padded frames carry no gradient)
    valid = (~preds[7]).unsqueeze(1).expand_as(x)          # the FFN works on (B, C, T)
    a = torch.where(valid, x.abs(), torch.full_like(x, 1e9)).reshape(-1)
    order = torch.argsort(a)[:4]
    flip = torch.zeros(a.shape, dtype=torch.bool)
    flip[order] = True
    state["flip"] = flip.reshape(x.shape)
    eps = float(a[order[-1]]) * 1.5
    g_flip, _ = f64_grad()
    monkeypatch.undo()
    monkeypatch.setattr(A, "RELU_BAND", eps)
    monkeypatch.setattr(A, "GATE_FLOOR", 1e-9)
    party = {"grads": g_flip, "mel": preds[0].detach().numpy(), "mel_post": preds[1].detach().numpy()}
    rep = setup["run"]({"engine": party})
    raw = max(r["engine"]["raw"] for r in rep["tensors"].values())
    print("REP", rep["parties"], rep.get("relu_candidates_in_band"), [float(a[i]) for i in order], {n: r["engine"] for n, r in rep["tensors"].items() if "explained" in r["engine"]})
    assert raw > 1e-6, raw                       # the flips moved something
    assert rep["parties"]["engine"]["relu_flips_used"] == 4, rep["parties"]
    assert rep["parties"]["engine"]["relu_units_priced"] >= 4
    for n, r in rep["tensors"].items():
        if "explained" in r["engine"]:
            assert r["engine"]["explained"] < 1e-9, (n, r)
    assert rep["pass"]
