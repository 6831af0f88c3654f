"""Reference-shaped host layer (model.py / systems.py / checkpoint.py): registry, hook names, 12-tuple / 10-tuple layouts,
state_dict + checkpoint key layout incl. the torch-Adam / LambdaLR state, loader surgery (system.py:115-192), 1-shot test mode,
avg_train_spk_emb, the Saver's result tree, Noam schedule.  Every test runs twice: through the SIMT emulator build on CPU and,
marked gpu, through libmtts.so on the MI355X (checkpoint import / export, optimizer-step restore and the test loop then move real
device state)."""
import json
import os

import numpy as np
import pytest
import torch

import __graft_entry__ as ge
from oracle_util import O, heads, synth, tiny_dims, torch_buffers
from meta_tts_amd import checkpoint
from meta_tts_amd.config import default_algorithm_config, default_train_config
from meta_tts_amd.systems import Trainer, get_system, noam_lr


@pytest.fixture(scope="module", params=[pytest.param("emu", id="emu"), pytest.param("gpu", id="gpu", marks=pytest.mark.gpu)])
def emu_lib(request):
    """Library the systems load: the emulator build, or None = meta_tts_amd/libmtts.so (the HIP build) on the GPU arm."""
    if request.param == "gpu":
        ge.build_device()
        return None
    return ge.build_emulator()


@pytest.fixture()
def cfgs(tmp_path):
    return make_cfgs(tmp_path)


def make_cfgs(tmp_path):
    dims = tiny_dims()
    pre = dims.preprocess_config
    pre["path"] = {"preprocessed_path": str(tmp_path)}
    pre["dataset"] = "LibriTTS"
    (tmp_path / "stats.json").write_text(json.dumps({"pitch": [-2.0, 8.0, 0.0, 1.0], "energy": [-1.5, 7.0, 0.0, 1.0]}))
    (tmp_path / "speakers.json").write_text(json.dumps({str(i): i for i in range(12)}))
    return pre, dims.model_config, default_train_config(), default_algorithm_config()


def _kw(n_mel):
    return dict(n_mel=n_mel, s_range=(5, 13), d_range=(1, 6), first_len=12)


def _system(cfgs, emu_lib, kind="meta", tasks=1):
    return get_system(kind)(*cfgs, max_tasks=tasks, max_batch=3, max_src_len=16, max_mel_len=96, lib_path=emu_lib)


def test_registry_and_schedule():
    assert get_system("meta").__name__ == "MetaSystem" and get_system("baseline").__name__ == "BaselineSystem"
    assert get_system("imaml").__name__ == "IMAMLSystem"      # lightning/systems/__init__.py:5-11 registers all three
    with pytest.raises(KeyError):
        get_system("anil")
    trn = default_train_config()
    for s in (0, 1, 3999, 4000, 300001):
        assert abs(noam_lr(s, 256, trn) - O.noam_lr(s)) < 1e-15


def test_forward_signature_and_tuple_layouts(cfgs, emu_lib):
    sysm = _system(cfgs, emu_lib)
    dims = sysm.model.dims
    assert dims.n_speaker == 12  # read from speakers.json
    b = synth.make_batch(3, 2, speaker=4, vocab=dims.vocab, **_kw(dims.n_mel))
    loss, out = sysm.common_step(tuple(torch.from_numpy(x) if isinstance(x, np.ndarray) else x for x in b), 0, train=True)
    assert len(out) == 10 and len(loss) == 6
    mel, mel_post, p, e, logd, d_rounded, src_masks, mel_masks, src_lens, mel_lens = out
    assert mel.shape == (2, b[8], dims.n_mel) and p.shape == (2, b[5]) and src_masks.dtype == torch.bool
    # against the oracle with the same (seed-0) parameters
    prm = {k: torch.from_numpy(v.copy()) for k, v in synth.make_params(dims, 0).items()}
    tb = O.to_torch_batch(b)
    with torch.no_grad():
        o = O.fs2_forward(prm, torch_buffers(dims), *tb[2:], n_head=heads(dims), max_seq_len=dims.max_seq_len, training=True)
        lo = O.fs2_loss(tb, o)
    assert (mel_post - o[1]).abs().max() < 5e-5
    np.testing.assert_allclose([float(x) for x in loss], [float(x) for x in lo], rtol=2e-5)
    assert torch.equal(src_masks, o[6]) and torch.equal(mel_masks, o[7])


def test_meta_training_step_and_checkpoint_roundtrip(cfgs, emu_lib, tmp_path):
    sysm = _system(cfgs, emu_lib)
    dims = sysm.model.dims
    sup = synth.make_batch(5, 3, speaker=2, vocab=dims.vocab, **_kw(dims.n_mel))
    qry = synth.make_batch(6, 2, speaker=2, vocab=dims.vocab, **_kw(dims.n_mel))
    out = sysm.training_step([([sup], [qry])], 0)
    assert set(out["log"]) == {f"Train/{k}" for k in ("Total Loss", "Mel Loss", "Mel-Postnet Loss", "Pitch Loss", "Energy Loss", "Duration Loss")}
    with pytest.raises(AssertionError):
        sysm.training_step([([sup], [qry]), ([sup], [qry])], 0)  # one task per process (base_adaptor.py:128)
    sysm.optimizer_step()
    sd = sysm.state_dict()
    assert "model.decoder.layer_stack.1.pos_ffn.w_1.weight" in sd and "learner.module.decoder.layer_stack.1.pos_ffn.w_1.weight" in sd
    assert "learner.module.encoder.src_word_emb.weight" not in sd  # the encoder is not adapted
    assert sd["model.encoder.position_enc"].shape == (1, dims.max_seq_len + 1, dims.d_model)
    assert "model.postnet.convolutions.0.1.running_mean" in sd
    path = str(tmp_path / "last.ckpt")
    checkpoint.save_checkpoint(sysm, path)
    raw = torch.load(path, map_location="cpu", weights_only=False)
    assert raw["global_step"] == 1 and "state_dict" in raw and "optimizer_states" in raw
    other = _system(cfgs, emu_lib)
    changes = checkpoint.load_checkpoint(other, path)
    assert changes == {"skip": [], "drop": [], "replace": [], "miss": []}
    for k, v in sysm.state_dict().items():
        np.testing.assert_array_equal(other.state_dict()[k], v)
    # resumed optimiser: the next identical step lands on identical parameters
    for s in (sysm, other):
        s.training_step([([sup], [qry])], 1)
        s.optimizer_step()
    np.testing.assert_array_equal(sysm.engine.export("mel_linear.weight"), other.engine.export("mel_linear.weight"))


def test_checkpoint_optimizer_state_is_torch_adam_layout(cfgs, emu_lib, tmp_path):
    """the file must be resumable by the reference — `optimizer.load_state_dict(ckpt["optimizer_states"][0])` on
    torch.optim.Adam over model.parameters() (lightning/optimizer.py:9-15) and LambdaLR.load_state_dict — and a checkpoint
    written by torch's Adam must resume here with the moments AND the bias-correction step count."""
    from meta_tts_amd.systems import noam_lr as host_noam
    sysm = _system(cfgs, emu_lib)
    dims = sysm.model.dims
    trn = cfgs[2]
    sup = synth.make_batch(5, 3, speaker=2, vocab=dims.vocab, **_kw(dims.n_mel))
    qry = synth.make_batch(6, 2, speaker=2, vocab=dims.vocab, **_kw(dims.n_mel))
    for step in range(2):
        sysm.training_step([([sup], [qry])], step)
        sysm.optimizer_step()
    path = str(tmp_path / "a.ckpt")
    checkpoint.save_checkpoint(sysm, path)
    raw = torch.load(path, map_location="cpu", weights_only=False)
    assert {"epoch", "global_step", "state_dict", "optimizer_states", "lr_schedulers"} <= set(raw)
    # --- the reference side: a torch Adam over parameters in model.parameters() order accepts the state -------------------
    spec = synth.param_spec(dims)
    tparams = [torch.nn.Parameter(raw["state_dict"]["model." + n].clone(), requires_grad=tr) for n, (_, tr) in spec.items()]
    o = trn["optimizer"]
    opt = torch.optim.Adam(tparams, lr=dims.d_model ** -0.5, betas=o["betas"], eps=o["eps"], weight_decay=o["weight_decay"])
    sch = torch.optim.lr_scheduler.LambdaLR(opt, lambda s: host_noam(s, dims.d_model, trn) / dims.d_model ** -0.5)
    opt.load_state_dict(raw["optimizer_states"][0])
    sch.load_state_dict(raw["lr_schedulers"][0])
    assert sch.last_epoch == 2 and abs(sch.get_last_lr()[0] - host_noam(2, dims.d_model, trn)) < 1e-12
    names = list(spec)
    i = names.index("mel_linear.weight")
    st = opt.state[tparams[i]]
    assert float(st["step"]) == 2.0
    np.testing.assert_array_equal(st["exp_avg"].numpy(), sysm.engine.export("mel_linear.weight", 4))
    assert names.index("encoder.position_enc") not in raw["optimizer_states"][0]["state"]  # frozen: in the group, no state
    # --- third step on both sides from the same gradient: torch Adam (after clip) == the engine's fused clip + Adam --------
    sysm.training_step([([sup], [qry])], 2)
    grads = {n: sysm.engine.export(n, 1) for n, (_, tr) in spec.items() if tr}
    for p_, (n, (_, tr)) in zip(tparams, spec.items()):
        p_.grad = torch.from_numpy(grads[n].copy()) if tr else None
    torch.nn.utils.clip_grad_norm_([p_ for p_ in tparams if p_.requires_grad], o["grad_clip_thresh"])
    opt.step(); sch.step()
    sysm.optimizer_step()
    for n in ("mel_linear.weight", "decoder.layer_stack.1.pos_ffn.w_1.weight", "postnet.convolutions.0.1.weight"):
        np.testing.assert_allclose(sysm.engine.export(n), tparams[names.index(n)].detach().numpy(), rtol=2e-5, atol=2e-7)
    # --- and back: a checkpoint whose optimizer state was written by torch resumes here ----------------------------------
    raw["optimizer_states"] = [opt.state_dict()]
    raw["lr_schedulers"] = [sch.state_dict()]
    raw["global_step"] = 3
    raw["state_dict"] = {k: (torch.tensor(v.item()) if np.ndim(v) == 0 else torch.from_numpy(np.ascontiguousarray(v))) for k, v in sysm.state_dict().items()}
    path2 = str(tmp_path / "b.ckpt")
    torch.save(raw, path2)
    other = _system(cfgs, emu_lib)
    checkpoint.load_checkpoint(other, path2)
    assert other.global_step == 3 and other.adam_steps == 3
    np.testing.assert_allclose(other.engine.export("mel_linear.weight", 5), opt.state[tparams[i]]["exp_avg_sq"].numpy(), rtol=1e-6)
    for s_ in (sysm, other):
        s_.training_step([([sup], [qry])], 3)
        s_.optimizer_step()
    np.testing.assert_allclose(other.engine.export("mel_linear.weight"), sysm.engine.export("mel_linear.weight"), rtol=1e-6, atol=1e-8)


def test_loader_surgery_old_key_and_speaker_table():
    """system.py:122-148: rename model.speaker_emb.weight; 326-row table -> 2390-row table keeps rows [:247] and [-79:]."""
    g = np.random.RandomState(0)
    model_sd = {"model.speaker_emb.model.weight": g.randn(2390, 4).astype(np.float32), "model.mel_linear.bias": np.zeros(3, np.float32)}
    old = {"model.speaker_emb.weight": g.randn(326, 4).astype(np.float32), "model.mel_linear.bias": np.ones(3, np.float32),
           "vocoder.mel2wav.x": np.zeros(2, np.float32)}
    sd, changes, changed = checkpoint.adapt_state_dict(old, model_sd, "LibriTTS")
    assert changed and changes["replace"] == [["model.speaker_emb.weight", "model.speaker_emb.model.weight"]]
    assert changes["drop"] == ["vocoder.mel2wav.x"]
    t = sd["model.speaker_emb.model.weight"]
    assert t.shape == (2390, 4)
    np.testing.assert_array_equal(t[:247], old["model.speaker_emb.weight"][:247])
    np.testing.assert_array_equal(t[-79:], old["model.speaker_emb.weight"][-79:])
    np.testing.assert_array_equal(t[247:-79], model_sd["model.speaker_emb.model.weight"][247:-79])


def test_baseline_step_matches_oracle_gradient(cfgs, emu_lib):
    sysm = _system(cfgs, emu_lib, kind="baseline")
    dims = sysm.model.dims
    b = synth.make_batch(8, 3, speaker=1, vocab=dims.vocab, **_kw(dims.n_mel))
    out = sysm.training_step(b, 0)
    prm = {k: torch.from_numpy(v.copy()) for k, v in synth.make_params(dims, 0).items()}
    prm["mel_linear.weight"].requires_grad_(True)
    tb = O.to_torch_batch(b)
    lo = O.fs2_loss(tb, O.fs2_forward(prm, torch_buffers(dims), *tb[2:], n_head=heads(dims), max_seq_len=dims.max_seq_len, training=True))
    g = torch.autograd.grad(lo[0], prm["mel_linear.weight"])[0].numpy()
    assert abs(out["loss"] - float(lo[0])) < 1e-4
    assert np.abs(sysm.engine.export("mel_linear.weight", 1) - g).max() < 1e-3 * np.abs(g).max()


def test_few_shot_test_step_matches_oracle(cfgs, emu_lib):
    """base_adaptor.py:155-189 — step 0 (eval), cumulative first-order adaptation in chunks, recon in train mode."""
    pre, mod, trn, alg = cfgs
    alg["adapt"]["train"]["steps"] = 2
    alg["adapt"]["test"]["steps"] = 4
    alg["adapt"]["test"]["saving_steps"] = [4]
    alg["adapt"]["task"]["lr"] = 0.01
    alg["adapt"]["train"]["lr"] = alg["adapt"]["test"]["lr"] = 0.01
    sysm = _system((pre, mod, trn, alg), emu_lib)
    dims = sysm.model.dims
    sup = synth.make_batch(5, 3, speaker=2, vocab=dims.vocab, **_kw(dims.n_mel))
    qry = synth.make_batch(6, 1, speaker=2, vocab=dims.vocab, **_kw(dims.n_mel))
    out = sysm.test_step([([sup], [qry])], 0)[0]
    assert set(out) == {"_batch", "step_0", "step_2", "step_4"}
    assert "synth" in out["step_0"] and "synth" in out["step_4"] and "synth" not in out["step_2"]
    # oracle
    prm = {k: torch.from_numpy(v.copy()) for k, v in synth.make_params(dims, 0).items()}
    for k, v in prm.items():
        if not k.endswith(("position_enc", "pitch_bins", "energy_bins")):
            v.requires_grad_(True)
    buf = torch_buffers(dims)
    ts, tq = O.to_torch_batch(sup), O.to_torch_batch(qry)
    kw = dict(n_head=heads(dims), max_seq_len=dims.max_seq_len)
    with torch.no_grad():
        o0 = O.fs2_forward(prm, buf, ts[2], *tq[3:], training=False, average_spk_emb=True, **kw)
        l0 = O.fs2_loss(tq, o0)
        s0 = O.fs2_forward(prm, buf, ts[2], *tq[3:6], training=False, average_spk_emb=True, **kw)
    np.testing.assert_allclose([float(x) for x in out["step_0"]["recon"]["losses"]], [float(x) for x in l0], rtol=5e-5)
    assert out["step_0"]["synth"]["output"][1].shape == s0[1].shape
    assert (out["step_0"]["synth"]["output"][1] - s0[1]).abs().max() < 5e-5
    names = O.adapted_names(prm, alg["adapt"]["modules"])
    fast = {k: prm[k] for k in names}
    for chunk in (2, 4):
        for _ in range(2):
            cur = dict(prm); cur.update(fast)
            l = O.fs2_loss(ts, O.fs2_forward(cur, buf, *ts[2:], training=True, **kw))
            g = torch.autograd.grad(l[0], [fast[k] for k in names])
            fast = {k: (fast[k] - 0.01 * gi).detach().requires_grad_(True) for k, gi in zip(names, g)}
        cur = dict(prm); cur.update(fast)
        with torch.no_grad():
            lq = O.fs2_loss(tq, O.fs2_forward(cur, buf, ts[2], *tq[3:], training=True, average_spk_emb=True, **kw))
        np.testing.assert_allclose([float(x) for x in out[f"step_{chunk}"]["recon"]["losses"]], [float(x) for x in lq], rtol=2e-4)


def test_one_shot_test_mode_adapts_on_each_support_utterance(cfgs, emu_lib):
    """base_adaptor.py:139-147 — `adapt.test.1-shot`: one _test_step per single support utterance (Task batch_size 1,
    no shuffle), all against the same query set; each equals a plain test_step on that 1-utterance support set."""
    from meta_tts_amd.data import split_reprocess
    pre, mod, trn, alg = cfgs
    alg["adapt"]["train"]["steps"] = 1
    alg["adapt"]["test"]["steps"] = 1
    alg["adapt"]["test"]["saving_steps"] = []
    alg["adapt"]["test"]["1-shot"] = True
    sysm = _system((pre, mod, trn, alg), emu_lib)
    dims = sysm.model.dims
    prm = synth.make_params(dims, 0)
    prm["variance_adaptor.duration_predictor.linear_layer.bias"][:] = np.log(3.0)   # random init predicts 0 frames
    sysm.engine.load_params(prm)
    sup = synth.make_batch(5, 3, speaker=2, vocab=dims.vocab, **_kw(dims.n_mel))
    qry = synth.make_batch(6, 2, speaker=2, vocab=dims.vocab, **_kw(dims.n_mel))
    outs = sysm.test_step([([sup], [qry])], 0)
    assert len(outs) == 3 and all(set(o) == {"_batch", "step_0", "step_1"} for o in outs)
    alg["adapt"]["test"]["1-shot"] = False
    for i, o in enumerate(outs):
        one = split_reprocess(sup, [i])
        assert one[3].shape == (1, int(sup[4][i])) and one[6].shape == (1, int(sup[7][i]), dims.n_mel) and one[0] == [sup[0][i]]
        ref = sysm.test_step([([one], [qry])], 0)[0]
        np.testing.assert_allclose([float(x) for x in o["step_1"]["recon"]["losses"]],
                                   [float(x) for x in ref["step_1"]["recon"]["losses"]], rtol=1e-6)
    l = [[float(x) for x in o["step_1"]["recon"]["losses"]] for o in outs]
    assert l[0] != l[1] and l[1] != l[2]      # different support utterances give different adapted models


def test_meta_system_trains_second_order_like_the_reference(cfgs, emu_lib):
    """base_adaptor.py:107 `first_order = not train`: MetaSystem's training step differentiates THROUGH the inner steps
    unless `adapt.first_order` says otherwise; both variants against the oracle's outer gradient."""
    pre, mod, trn, alg = cfgs
    alg["adapt"]["train"]["steps"] = 2
    alg["adapt"]["task"]["lr"] = 0.001
    got = {}
    for fo in (None, True):
        if fo is None:
            alg["adapt"].pop("first_order", None)
        else:
            alg["adapt"]["first_order"] = fo
        sysm = _system((pre, mod, trn, alg), emu_lib)
        dims = sysm.model.dims
        sup = synth.make_batch(5, 3, speaker=2, vocab=dims.vocab, **_kw(dims.n_mel))
        qry = synth.make_batch(6, 2, speaker=2, vocab=dims.vocab, **_kw(dims.n_mel))
        sysm.training_step([([sup], [qry])], 0)
        got[fo] = {n: sysm.engine.export(n, 1) for n in ("mel_linear.weight", "encoder.layer_stack.0.slf_attn.fc.weight")}
    for fo, second in ((None, True), (True, False)):
        prm = {k: torch.from_numpy(v.copy()) for k, v in synth.make_params(dims, 0).items()}
        for k, v in prm.items():
            if not k.endswith(("position_enc", "pitch_bins", "energy_bins")):
                v.requires_grad_(True)
        ql, _, _, _ = O.maml_task(prm, torch_buffers(dims), O.to_torch_batch(sup), O.to_torch_batch(qry), steps=2, lr=0.001,
                                  second_order=second, modules=alg["adapt"]["modules"], n_head=heads(dims), max_seq_len=dims.max_seq_len)
        for n, x in got[fo].items():
            ref = torch.autograd.grad(ql[0], prm[n], retain_graph=True)[0].numpy()
            assert np.abs(x - ref).max() <= 2e-3 * np.abs(ref).max() + 1e-7, (fo, n)
    assert np.abs(got[None]["encoder.layer_stack.0.slf_attn.fc.weight"] - got[True]["encoder.layer_stack.0.slf_attn.fc.weight"]).max() > 0


def test_on_test_start_avg_train_spk_emb(cfgs, emu_lib, tmp_path):
    """system.py:194-212: the 39 unseen LibriTTS test speakers start from the mean of the 247 train-clean-100 rows."""
    import json
    from meta_tts_amd.config import SYNTH_STATS
    pre, mod, trn, alg = cfgs
    with open(tmp_path / "speakers.json", "w") as f:
        json.dump({f"s{i}": i for i in range(300)}, f)
    with open(tmp_path / "stats.json", "w") as f:
        json.dump(SYNTH_STATS, f)
    pre = dict(pre)
    pre["path"] = {"preprocessed_path": str(tmp_path)}
    pre["dataset"] = "LibriTTS"
    alg["adapt"]["test"]["avg_train_spk_emb"] = True
    sysm = _system((pre, mod, trn, alg), emu_lib)
    assert sysm.model.dims.n_speaker == 300
    g = np.random.RandomState(0)
    w = g.standard_normal((300, sysm.model.dims.d_model)).astype(np.float32)
    sysm.engine.load_params({"speaker_emb.model.weight": w}, strict=False)
    sysm.on_test_start()
    after = sysm.engine.export("speaker_emb.model.weight")
    np.testing.assert_array_equal(after[:-39], w[:-39])
    np.testing.assert_allclose(after[-39:], np.broadcast_to(w[:247].mean(axis=0), (39, w.shape[1])), rtol=1e-6, atol=1e-7)
    alg["adapt"]["test"]["avg_train_spk_emb"] = False
    sysm.engine.load_params({"speaker_emb.model.weight": w}, strict=False)
    sysm.on_test_start()
    np.testing.assert_array_equal(sysm.engine.export("speaker_emb.model.weight"), w)


def test_bins_follow_the_checkpoint_and_stale_predictions_are_refused(cfgs, emu_lib):
    """pitch / energy bins are frozen nn.Parameters the reference restores from the checkpoint
    (modules.py:57-71) — a cross-corpus checkpoint must change the quantisation; and FastSpeech2Loss must not silently
    evaluate somebody else's activations."""
    from meta_tts_amd.engine import MttsError
    sysm = _system(cfgs, emu_lib)
    dims = sysm.model.dims
    b = synth.make_batch(3, 2, speaker=4, vocab=dims.vocab, **_kw(dims.n_mel))
    tb = tuple(torch.from_numpy(x) if isinstance(x, np.ndarray) else x for x in b)
    sd = sysm.model.state_dict()
    np.testing.assert_allclose(sd["variance_adaptor.pitch_bins"], np.linspace(dims.pitch_min, dims.pitch_max, dims.n_bins - 1), rtol=1e-6)
    loss0, out0 = sysm.common_step(tb, 0, train=False)
    # a checkpoint from a corpus with a different pitch range
    sd2 = dict(sd)
    sd2["variance_adaptor.pitch_bins"] = np.linspace(-0.5, 0.5, dims.n_bins - 1).astype(np.float32)
    sysm.model.load_state_dict(sd2)
    np.testing.assert_array_equal(sysm.model.state_dict()["variance_adaptor.pitch_bins"], sd2["variance_adaptor.pitch_bins"])
    loss1, out1 = sysm.common_step(tb, 0, train=False)
    prm = {k: torch.from_numpy(np.asarray(v).copy()) for k, v in sd2.items() if "running" not in k and "num_batches" not in k}
    ob = O.to_torch_batch(b)
    with torch.no_grad():
        o = O.fs2_forward(prm, torch_buffers(dims), *ob[2:], n_head=heads(dims), max_seq_len=dims.max_seq_len, training=False)
    assert (out1[1] - o[1]).abs().max() < 5e-5           # the engine quantises with the checkpoint's bins
    assert (out1[1] - out0[1]).abs().max() > 1e-3        # ... which is not what the local stats.json would give
    with pytest.raises(MttsError, match="non-decreasing"):
        sysm.engine.set_bins(pitch_bins=np.linspace(1, -1, dims.n_bins - 1))
    # stale predictions: out0 was produced two forwards ago
    with pytest.raises(MttsError, match="stale"):
        sysm.loss_func(tb, out0)
    assert len(sysm.loss_func(tb, out1)) == 6


def test_test_stage_result_layout_matches_reference_saver(cfgs, emu_lib, tmp_path):
    """saver.py:130-178 / :203-213 + callbacks/utils.py:55-141: the result tree `evaluation/` globs
    (`audio/Testing/step_<S>/<task>/<id>.recon.wav`, `...FTstep_<k>.synth.wav`, `csv/Testing/step_<S>/<task>.csv`), with the CSV
    bytes equal to what pandas writes for the same values (the reference's `DataFrame.to_csv`)."""
    pd = pytest.importorskip("pandas")
    from scipy.io import wavfile
    from meta_tts_amd.saver import CSV_COLUMNS, Saver, loss2dict
    pre, mod, trn, alg = cfgs
    alg["adapt"]["train"]["steps"] = 2
    alg["adapt"]["test"]["steps"] = 4
    alg["adapt"]["test"]["saving_steps"] = [4]
    alg["adapt"]["task"]["lr"] = 0.01
    sysm = _system((pre, mod, trn, alg), emu_lib)
    dims = sysm.model.dims
    prm = synth.make_params(dims, 0)
    prm["variance_adaptor.duration_predictor.linear_layer.bias"][:] = 1.2   # a few predicted frames at random init
    sysm.model.load_state_dict(prm)
    sup = synth.make_batch(5, 3, speaker=2, vocab=dims.vocab, **_kw(dims.n_mel))
    qry = synth.make_batch(6, 2, speaker=2, vocab=dims.vocab, **_kw(dims.n_mel))
    batch = [([sup], [qry])]
    sysm.test_global_step = 100000
    outs = sysm.test_step(batch, 0)

    class StubVocoder:   # LightningMelGAN.infer contract (lightning/utils.py:20-30): list of int16 arrays cropped to `lengths`
        def infer(self, mels, max_wav_value, lengths=None):
            return [(np.tanh(m.mean(axis=0)).repeat(256)[:l] * 0.5 * max_wav_value).astype("int16") for m, l in zip(mels, lengths)]

    sq = {f"{'-'.join(sup[0])}.{'-'.join(qry[0])}": "test_007"}
    sv = Saver(pre, str(tmp_path / "log"), str(tmp_path / "result"))
    paths = sv.on_test_batch_end(outs, batch, sq, sysm.test_global_step, sysm.adaptation_steps, sysm.test_adaptation_steps, StubVocoder())
    root = tmp_path / "result"
    csv_path = root / "csv" / "Testing" / "step_100000" / "test_007.csv"
    assert paths == [str(csv_path)]
    # pandas on the same loss values, exactly as saver.py:174-175 does
    rows = [{"Step": k, **loss2dict(outs[0][f"step_{k}"]["recon"]["losses"])} for k in (0, 2, 4)]
    ref = tmp_path / "ref.csv"
    pd.DataFrame(rows, columns=["Step"] + CSV_COLUMNS).set_index("Step").to_csv(ref, mode="a", header=True, index=True)
    assert csv_path.read_bytes() == ref.read_bytes()
    adir = root / "audio" / "Testing" / "step_100000" / "test_007"
    names = sorted(os.listdir(adir))
    want = sorted([f"{i}.recon.wav" for i in qry[0]] + [f"{i}.step_100000-FTstep_{k}.synth.wav" for i in qry[0] for k in (0, 4)])
    assert names == want
    rate, wav = wavfile.read(adir / f"{qry[0][0]}.recon.wav")
    assert rate == 22050 and wav.dtype == np.int16 and len(wav) == int(qry[7][0]) * 256
    rate, wav = wavfile.read(adir / f"{qry[0][0]}.step_100000-FTstep_4.synth.wav")
    assert len(wav) == int(outs[0]["step_4"]["synth"]["output"][9][0]) * 256
    assert (root / "figure" / "Testing" / "step_100000" / "test_007").is_dir()
    # validation CSV: one appended row per pass, header once (saver.py:203-213)
    v = sysm.validation_step(batch, 0)
    p1 = sv.on_validation_batch_end(v, batch, global_step=999, val_SQids2Tid={k: "val_003" for k in sq})
    sv.on_validation_batch_end(v, batch, global_step=1999, val_SQids2Tid={k: "val_003" for k in sq})
    ref2 = tmp_path / "ref2.csv"
    for step in (1000, 2000):
        pd.DataFrame(loss2dict(v["losses"]), columns=CSV_COLUMNS, index=[step]).to_csv(ref2, mode="a", header=not os.path.exists(ref2), index=True, index_label="Step")
    assert open(p1, "rb").read() == ref2.read_bytes() and p1.endswith(os.path.join("log", "csv", "Validation", "val_003.csv"))
    # 1-shot mode: several outputs per task -> task ids suffixed _<i> (saver.py:143-147)
    alg["adapt"]["test"]["1-shot"] = True
    one = sysm.test_step(batch, 0)
    assert len(one) == 3
    paths = sv.on_test_batch_end(one, batch, sq, 100000, sysm.adaptation_steps, sysm.test_adaptation_steps, None)
    assert [os.path.basename(p) for p in paths] == ["test_007_0.csv", "test_007_1.csv", "test_007_2.csv"]



def test_imaml_system_training_step_and_test_loop(cfgs, emu_lib):
    """IMAMLSystem (imaml.py:22-195) through the registry: the `adapt.imaml` block of config/algorithm/dev.yaml:22-26, one training
    step (mini-batch proximal inner loop + CG hypergradient + manual optimiser step without a second clip), validation, test loop."""
    pre, mod, trn, alg = cfgs
    alg["type"] = "imaml"
    alg["adapt"]["imaml"] = {"K": 2, "reg_param": 1.0, "batch_size": 2, "stochastic": True}
    alg["adapt"]["train"]["steps"] = 2
    alg["adapt"]["test"]["steps"] = 4
    alg["adapt"]["task"]["lr"] = 0.02
    sysm = _system((pre, mod, trn, alg), emu_lib, kind="imaml")
    dims = sysm.model.dims
    prm = synth.make_params(dims, 0)
    prm["variance_adaptor.duration_predictor.linear_layer.bias"][:] = 1.2   # some predicted frames for the step-0 synthesis
    sysm.model.load_state_dict(prm)
    sup = synth.make_batch(5, 3, speaker=2, vocab=dims.vocab, **_kw(dims.n_mel))
    qry = synth.make_batch(6, 2, speaker=2, vocab=dims.vocab, **_kw(dims.n_mel))
    batch = [([sup], [qry])]
    before = sysm.engine.export("mel_linear.weight")
    enc_before = sysm.engine.export("encoder.layer_stack.0.slf_attn.fc.weight")
    out = sysm.training_step(batch, 0)
    assert np.isfinite(out["losses"]).all() and set(out["log"]) == {f"Train/{k}" for k in LOSS_KEYS}
    g = sysm.engine.export("mel_linear.weight", 1)
    assert np.abs(g).max() > 0 and not sysm.engine.export("encoder.layer_stack.0.slf_attn.fc.weight", 1).any()
    assert float(np.sqrt(sum((sysm.engine.export(n, 1).astype(np.float64) ** 2).sum() for n in sysm.engine.params))) <= 1.0 + 1e-4  # per-task clip
    sysm.optimizer_step()
    assert np.abs(sysm.engine.export("mel_linear.weight") - before).max() > 0
    np.testing.assert_array_equal(sysm.engine.export("encoder.layer_stack.0.slf_attn.fc.weight"), enc_before)  # zero hypergradient, Adam leaves it
    v = sysm.validation_step(batch, 0)
    assert np.isfinite(v["losses"]).all()
    outs = sysm.test_step(batch, 0)
    assert set(outs[0]) == {"_batch", "step_0", "step_2", "step_4"}
    l0, l4 = float(outs[0]["step_0"]["recon"]["losses"][0]), float(outs[0]["step_4"]["recon"]["losses"][0])
    assert np.isfinite([l0, l4]).all() and l4 != l0


LOSS_KEYS = ("Total Loss", "Mel Loss", "Mel-Postnet Loss", "Pitch Loss", "Energy Loss", "Duration Loss")


def test_shared_speaker_embedding_mode(cfgs, emu_lib):
    """adapt.speaker_emb: shared (config/algorithm/*share_emb*.yaml; speaker_encoder.py:52-53,67-69): nn.Embedding(1, d) indexed with
    zeros_like(speaker ids) — every utterance of every speaker reads and trains the same row."""
    pre, mod, trn, alg = cfgs
    alg["adapt"]["speaker_emb"] = "shared"
    sysm = _system((pre, mod, trn, alg), emu_lib)
    dims = sysm.model.dims
    assert dims.n_speaker == 1 and sysm.model.state_dict()["speaker_emb.model.weight"].shape == (1, dims.d_model)
    sup = synth.make_batch(5, 3, speaker=7, vocab=dims.vocab, **_kw(dims.n_mel))     # ids 7 / 9 would be out of range for a 1-row table
    qry = synth.make_batch(6, 2, speaker=9, vocab=dims.vocab, **_kw(dims.n_mel))
    q, s = sysm.meta_learn_tasks([(sup, qry)], train=False)
    prm = {k: torch.from_numpy(v.copy()) for k, v in synth.make_params(dims, 0).items()}
    for k, v in prm.items():
        if not k.endswith(("position_enc", "pitch_bins", "energy_bins")):
            v.requires_grad_(True)
    z = lambda b: tuple(b[:2]) + (np.zeros_like(b[2]),) + tuple(b[3:])
    mods = alg["adapt"]["modules"]
    ql, sl, fast, _ = O.maml_task(prm, torch_buffers(dims), O.to_torch_batch(z(sup)), O.to_torch_batch(z(qry)), steps=sysm.adaptation_steps,
                                  lr=sysm.adaptation_lr, second_order=False, modules=mods, n_head=heads(dims), max_seq_len=dims.max_seq_len)
    np.testing.assert_allclose(q[0], [float(x) for x in ql], rtol=1e-4)
    g = torch.autograd.grad(ql[0], prm["speaker_emb.model.weight"])[0].numpy()
    got = sysm.engine.export("speaker_emb.model.weight", 1)
    assert got.shape == (1, dims.d_model) and np.abs(got - g).max() <= 2e-3 * np.abs(g).max()
    alg["adapt"]["speaker_emb"] = "ge2e"     # not one of the reference's modes
    with pytest.raises(Exception, match="speaker_emb"):
        _system((pre, mod, trn, alg), emu_lib)


def test_dvec_speaker_mode_baseline_system(cfgs, emu_lib):
    """config/algorithm/dvec.yaml: `type: baseline`, `speaker_emb: dvec`, `modules: []` — batch[2] is (ref_mels, ref_slices)
    (collate.py:29-43) and the frozen d-vector encoder (speaker_encoder.py:56-58,71-76) supplies the speaker embedding.  The training
    step equals the oracle's forward/backward with the oracle encoder's embeddings in place of the table rows."""
    from oracle import dvector_oracle as dvo
    pre, mod, trn, alg = cfgs
    alg["type"] = "baseline"
    alg["adapt"]["speaker_emb"] = "dvec"
    alg["adapt"]["modules"] = []
    dv = dict(n_mels=8, hidden=64, layers=2, frames=6)
    alg["adapt"]["dvector"] = dv
    sysm = _system((pre, mod, trn, alg), emu_lib, kind="baseline")
    dims = sysm.model.dims
    sd = sysm.model.state_dict()
    assert "speaker_emb.model.weight" not in sd and sd["speaker_emb.model.lstm.weight_hh_l1"].shape == (256, 64)
    assert sd["speaker_emb.model.linear.weight"].shape == (dims.d_model, 64)
    b = list(synth.make_batch(8, 3, speaker=1, vocab=dims.vocab, **_kw(dims.n_mel)))
    g = np.random.RandomState(2)
    counts = [2, 1, 3]
    ref_mels = g.standard_normal((sum(counts), dv["frames"], dv["n_mels"])).astype(np.float32)
    off = np.concatenate([[0], np.cumsum(counts)])
    slices = [slice(int(off[i]), int(off[i + 1])) for i in range(3)]
    b[2] = (ref_mels, slices)
    out = sysm.training_step(tuple(b), 0)
    enc_sd = {k[len("speaker_emb.model."):]: v for k, v in sd.items() if k.startswith("speaker_emb.model.")}
    emb = dvo.speaker_embeds(enc_sd, ref_mels, slices, n_mels=8, hidden=64, emb=dims.d_model, layers=2)
    prm = {k: torch.from_numpy(v.copy()) for k, v in synth.make_params(dims, 0).items()}
    prm["speaker_emb.model.weight"] = emb
    prm["mel_linear.weight"].requires_grad_(True)
    tb = list(O.to_torch_batch(tuple(b[:2]) + (np.arange(3),) + tuple(b[3:])))
    lo = O.fs2_loss(tuple(tb), O.fs2_forward(prm, torch_buffers(dims), *tb[2:], n_head=heads(dims), max_seq_len=dims.max_seq_len, training=True))
    gr = torch.autograd.grad(lo[0], prm["mel_linear.weight"])[0].numpy()
    assert abs(out["loss"] - float(lo[0])) < 1e-4
    assert np.abs(sysm.engine.export("mel_linear.weight", 1) - gr).max() < 1e-3 * np.abs(gr).max()
    # a checkpoint round trip keeps the encoder's tensors under the reference's names
    sysm.model.load_state_dict(sd)
    np.testing.assert_array_equal(sysm.model.state_dict()["speaker_emb.model.linear.bias"], sd["speaker_emb.model.linear.bias"])


def test_trained_speaker_encoder_baseline_step_and_checkpoint(cfgs, emu_lib, tmp_path):
    """config/algorithm/scratch_encoder.yaml (`type: baseline`, `speaker_emb: scratch_encoder`): the LSTM speaker encoder is trained
    with the acoustic model — one training step + optimizer step against torch (oracle forward, autograd through nn.LSTM, joint
    clip_grad_norm_, Adam), then the checkpoint keeps the encoder's Adam moments at torch's parameter positions."""
    from oracle import dvector_oracle as dvo
    from meta_tts_amd.checkpoint import load_checkpoint, save_checkpoint
    pre, mod, trn, alg = cfgs
    alg["type"] = "baseline"
    alg["adapt"]["speaker_emb"] = "scratch_encoder"
    alg["adapt"]["modules"] = []
    dv = dict(n_mels=8, hidden=64, layers=2, frames=6)
    alg["adapt"]["dvector"] = dv
    sysm = _system((pre, mod, trn, alg), emu_lib, kind="baseline")
    dims = sysm.model.dims
    kw = dict(n_mels=8, hidden=64, emb=dims.d_model, layers=2)
    sd0 = sysm.model.state_dict()
    enc_sd = {k[len("speaker_emb.model."):]: v.copy() for k, v in sd0.items() if k.startswith("speaker_emb.model.")}
    b = list(synth.make_batch(8, 3, speaker=1, vocab=dims.vocab, **_kw(dims.n_mel)))
    g = np.random.RandomState(3)
    counts = [1, 3, 2]
    ref_mels = g.standard_normal((sum(counts), dv["frames"], dv["n_mels"])).astype(np.float32)
    off = np.concatenate([[0], np.cumsum(counts)])
    slices = [slice(int(off[i]), int(off[i + 1])) for i in range(3)]
    b[2] = (ref_mels, slices)
    out = sysm.training_step(tuple(b), 0)
    # torch side: encoder (autograd) -> embeddings as the "table" rows of the oracle model -> loss
    lstm, linear = dvo.build(enc_sd, **kw)
    _, (hidden, _) = lstm(torch.from_numpy(ref_mels))
    raw = torch.relu(linear(hidden[-1]))
    pe = raw / torch.norm(raw, dim=1, keepdim=True)
    emb = torch.stack([torch.nn.functional.normalize(pe[sl].mean(dim=0), dim=0) for sl in slices])
    prm = {k: torch.from_numpy(v.copy()) for k, v in synth.make_params(dims, 0).items()}
    names = [k for k in prm if not k.endswith(("position_enc", "pitch_bins", "energy_bins")) and k != "speaker_emb.model.weight"]
    for k in names:
        prm[k].requires_grad_(True)
    prm["speaker_emb.model.weight"] = emb
    tb = list(O.to_torch_batch(tuple(b[:2]) + (np.arange(3),) + tuple(b[3:])))
    lo = O.fs2_loss(tuple(tb), O.fs2_forward(prm, torch_buffers(dims), *tb[2:], n_head=heads(dims), max_seq_len=dims.max_seq_len, training=True))
    assert abs(out["loss"] - float(lo[0])) < 1e-4
    enc_params = list(lstm.parameters()) + list(linear.parameters())
    lo[0].backward()
    enc = sysm.model.speaker_encoder
    for (n, p) in [(f"lstm.{n}", p) for n, p in lstm.named_parameters()] + [(f"linear.{n}", p) for n, p in linear.named_parameters()]:
        r = p.grad.numpy()
        assert np.abs(enc.export(n, 1) - r).max() <= 2e-3 * np.abs(r).max() + 1e-8, n
    # joint clip + Adam
    allp = [prm[k] for k in names] + enc_params
    total = float(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in allp if p.grad is not None)))
    o = trn["optimizer"]
    lr = sysm.optimizer_step()
    torch.nn.utils.clip_grad_norm_([p for p in allp if p.grad is not None], o["grad_clip_thresh"])
    opt = torch.optim.Adam([p for p in allp if p.grad is not None], lr=lr, betas=tuple(o["betas"]), eps=o["eps"], weight_decay=o["weight_decay"])
    opt.step()
    assert total > o["grad_clip_thresh"]          # the clip is active: the coefficient depends on BOTH parameter sets
    # first Adam step = lr * sign(g) for |g| >> eps: compare where the clipped gradient is well above eps
    w_new, w_ref, gr = enc.export("linear.weight"), linear.weight.detach().numpy(), linear.weight.grad.numpy()
    big = np.abs(gr) > 1e-6
    np.testing.assert_allclose(w_new[big], w_ref[big], rtol=0, atol=2e-3 * lr + 1e-7)
    m_new, m_ref = sysm.engine.export("mel_linear.weight"), prm["mel_linear.weight"].detach().numpy()
    bigm = np.abs(prm["mel_linear.weight"].grad.numpy()) > 1e-6
    np.testing.assert_allclose(m_new[bigm], m_ref[bigm], rtol=0, atol=2e-3 * lr + 1e-7)
    # the Adam moment of the encoder equals (1 - beta1) * clipped gradient
    coef = min(1.0, o["grad_clip_thresh"] / (total + 1e-6))
    np.testing.assert_allclose(enc.export("linear.weight", 2), (1 - o["betas"][0]) * coef * enc.export("linear.weight", 1), rtol=1e-4, atol=1e-9)
    # checkpoint: the encoder's tensors replace the table in the parameter order, its moments sit at their torch positions
    path = str(tmp_path / "enc.ckpt")
    save_checkpoint(sysm, path)
    ck = torch.load(path, map_location="cpu", weights_only=False)
    n_params = len(ck["optimizer_states"][0]["param_groups"][0]["params"])
    st = ck["optimizer_states"][0]["state"]
    assert "model.speaker_emb.model.lstm.weight_ih_l0" in ck["state_dict"] and "model.speaker_emb.model.weight" not in ck["state_dict"]
    np.testing.assert_allclose(st[n_params - 1]["exp_avg"].numpy(), enc.export("linear.bias", 2))      # linear.bias is the last parameter
    np.testing.assert_allclose(st[n_params - 2]["exp_avg_sq"].numpy(), enc.export("linear.weight", 3))
    sys2 = _system((pre, mod, trn, alg), emu_lib, kind="baseline")
    load_checkpoint(sys2, path)
    np.testing.assert_array_equal(sys2.model.speaker_encoder.export("lstm.weight_hh_l1"), enc.export("lstm.weight_hh_l1"))
    np.testing.assert_array_equal(sys2.model.speaker_encoder.export("lstm.weight_hh_l1", 3), enc.export("lstm.weight_hh_l1", 3))
    assert sys2.adam_steps == 1
    # a MAML system cannot train the encoder
    alg["type"] = "meta"
    with pytest.raises(Exception, match="baseline"):
        _system((pre, mod, trn, alg), emu_lib, kind="meta")


def test_dvec_few_shot_test_step_averages_support_embeddings(cfgs, emu_lib):
    """base_adaptor.py:64-67 with an encoder speaker mode: the query pass of `_test_step` uses the MEAN of the support utterances'
    d-vectors (speaker_args of the support batch, average_spk_emb=True), expanded to the query batch."""
    from oracle import dvector_oracle as dvo
    pre, mod, trn, alg = cfgs
    alg["type"] = "baseline"
    alg["adapt"]["speaker_emb"] = "dvec"
    alg["adapt"]["modules"] = []
    alg["adapt"]["test"]["steps"] = 0
    dv = dict(n_mels=8, hidden=64, layers=2, frames=6)
    alg["adapt"]["dvector"] = dv
    sysm = _system((pre, mod, trn, alg), emu_lib, kind="baseline")
    dims = sysm.model.dims
    g = np.random.RandomState(9)

    def with_refs(b, counts):
        mels = g.standard_normal((sum(counts), dv["frames"], dv["n_mels"])).astype(np.float32)
        off = np.concatenate([[0], np.cumsum(counts)])
        return tuple(b[:2]) + ((mels, [slice(int(off[i]), int(off[i + 1])) for i in range(len(counts))]),) + tuple(b[3:])
    sup = with_refs(synth.make_batch(5, 3, speaker=1, vocab=dims.vocab, **_kw(dims.n_mel)), [2, 1, 2])
    qry = with_refs(synth.make_batch(6, 2, speaker=1, vocab=dims.vocab, **_kw(dims.n_mel)), [1, 1])
    outs = sysm.test_step([((sup,), (qry,))], 0)
    got = float(outs[0]["step_0"]["recon"]["losses"][0])
    sd = sysm.model.state_dict()
    enc_sd = {k[len("speaker_emb.model."):]: v for k, v in sd.items() if k.startswith("speaker_emb.model.")}
    emb = dvo.speaker_embeds(enc_sd, sup[2][0], sup[2][1], n_mels=8, hidden=64, emb=dims.d_model, layers=2).mean(dim=0, keepdim=True).expand(2, -1)
    prm = {k: torch.from_numpy(v.copy()) for k, v in synth.make_params(dims, 0).items()}
    prm["speaker_emb.model.weight"] = emb
    tb = list(O.to_torch_batch(tuple(qry[:2]) + (np.arange(2),) + tuple(qry[3:])))
    with torch.no_grad():
        lo = O.fs2_loss(tuple(tb), O.fs2_forward(prm, torch_buffers(dims), *tb[2:], n_head=heads(dims), max_seq_len=dims.max_seq_len, training=False))
    assert abs(got - float(lo[0])) < 2e-4 * max(1.0, abs(float(lo[0])))


def test_trainer_refuses_an_exchange_view_without_the_tail(cfgs, emu_lib):
    """the collective covers the outer gradient AND the exchange tail (loss scalars + BatchNorm buffers); a view of the gradient alone
    would make every non-zero rank unpack its own zeroed tail into its BatchNorm running buffers.  Trainer must refuse it."""
    import torch
    pre, mc, tc, ac = cfgs
    sysm = get_system("meta")(pre, mc, tc, ac, max_tasks=1, max_batch=3, max_src_len=16, max_mel_len=96, lib_path=emu_lib)
    eng = sysm.engine
    assert eng.sync_floats > eng.n_total
    with pytest.raises(ValueError):
        Trainer(sysm, outer_grad_tensor=torch.zeros(eng.n_total))
    Trainer(sysm, outer_grad_tensor=torch.zeros(eng.sync_floats))      # the full exchange buffer is accepted
    assert eng.bn_pack_weight(0, 4) == 1.0 and eng.bn_pack_weight(2, 4) == 0.0
    eng.set_bn_sync("mean")
    assert eng.bn_pack_weight(2, 4) == 0.25


def test_gradient_accumulation_matches_one_step_on_the_joint_batch(cfgs, emu_lib):
    """optimizer.grad_acc_step = 2 (main.py:62 accumulate_grad_batches): two meta-batches of one task each, ONE optimizer step on the sum
    of their halved gradients — the same parameters as one step on a meta-batch holding both tasks; the optimizer must not move between
    the two calls.  iMAML and the trained speaker encoders refuse the combination instead of ignoring it."""
    pre, mc, tc, ac = cfgs
    import copy
    tc2 = copy.deepcopy(tc)
    tc2["optimizer"]["grad_acc_step"] = 2
    ac = copy.deepcopy(ac)
    ac["adapt"]["first_order"] = True
    dims = tiny_dims()
    kw = _kw(dims.n_mel)
    t1 = (synth.make_batch(5, 3, speaker=2, vocab=dims.vocab, **kw), synth.make_batch(6, 2, speaker=2, vocab=dims.vocab, **kw))
    t2 = (synth.make_batch(7, 2, speaker=4, vocab=dims.vocab, **kw), synth.make_batch(8, 3, speaker=4, vocab=dims.vocab, **kw))
    acc = get_system("meta")(pre, mc, tc2, ac, max_tasks=2, max_batch=3, max_src_len=16, max_mel_len=96, lib_path=emu_lib)
    tr = Trainer(acc)
    assert tr.grad_acc == 2
    before = acc.state_dict()["model.mel_linear.weight"].copy()
    _, _, lr = tr.meta_step([t1], total_tasks=1)
    assert lr is None and acc.global_step == 0
    np.testing.assert_array_equal(acc.state_dict()["model.mel_linear.weight"], before)   # nothing stepped yet
    q2, _, lr = tr.meta_step([t2], total_tasks=1)
    assert lr is not None and acc.global_step == 1
    # the reduced losses are those of the window's LAST micro-batch (each training_step logs its own, meta.py:78-79), not divided by N
    np.testing.assert_allclose(tr.synced_losses(), np.asarray(q2)[0], rtol=1e-6)
    ref = get_system("meta")(pre, mc, tc, ac, max_tasks=2, max_batch=3, max_src_len=16, max_mel_len=96, lib_path=emu_lib)
    Trainer(ref).meta_step([t1, t2], total_tasks=2)
    for k in ("model.mel_linear.weight", "model.decoder.layer_stack.1.pos_ffn.w_1.weight", "model.encoder.layer_stack.0.slf_attn.fc.weight",
              "model.postnet.convolutions.2.0.conv.bias"):
        a, b = acc.state_dict()[k], ref.state_dict()[k]
        np.testing.assert_allclose(a, b, rtol=1e-4, atol=1e-7, err_msg=k)
        assert np.abs(a - (before if k == "model.mel_linear.weight" else a * 0 + 1e9)).max() > 0
    # the accumulate flag is armed for ONE call: a direct gradient call between two batches of a window overwrites the outer buffer
    # instead of adding to the window's, and a failing call leaves the window where it was
    tr2 = Trainer(get_system("meta")(pre, mc, tc2, ac, max_tasks=2, max_batch=3, max_src_len=16, max_mel_len=96, lib_path=emu_lib))
    tr2.meta_step([t1], total_tasks=1)
    _, _, lr = tr2.meta_step([t2], total_tasks=1)          # second batch of the window: accumulates, steps
    assert lr is not None and tr2._acc_i == 0
    tr2.meta_step([t1], total_tasks=1)                     # a new window is open (outer buffer holds t1's gradient) ...
    tr2.system.meta_learn_tasks([t2], total_tasks=1)       # ... a direct call must OVERWRITE it
    direct = tr2.system.engine.export("mel_linear.weight", 1)
    fresh = get_system("meta")(pre, mc, tc, ac, max_tasks=2, max_batch=3, max_src_len=16, max_mel_len=96, lib_path=emu_lib)
    fresh.engine.load_params({k[len("model."):]: v for k, v in tr2.system.state_dict().items() if k.startswith("model.") and k[len("model."):] in fresh.engine.params})
    fresh.meta_learn_tasks([t2], total_tasks=1)
    np.testing.assert_allclose(direct, fresh.engine.export("mel_linear.weight", 1), rtol=1e-5, atol=1e-8)
    assert tr2._acc_i == 1
    with pytest.raises(Exception):
        tr2.meta_step([(t1[0],)], total_tasks=1)           # malformed task: the gradient call raises ...
    assert tr2._acc_i == 1                                  # ... and the window has not advanced
    tc0 = copy.deepcopy(tc)
    tc0["optimizer"]["grad_acc_step"] = 0
    with pytest.raises(ValueError):
        Trainer(get_system("meta")(pre, mc, tc0, ac, max_tasks=1, max_batch=3, max_src_len=16, max_mel_len=96, lib_path=emu_lib))
    alg = default_algorithm_config()
    alg["type"] = "imaml"
    alg["adapt"]["imaml"] = {"K": 2, "reg_param": 1.0, "batch_size": 2, "stochastic": True}
    im = get_system("imaml")(pre, mc, tc2, alg, max_tasks=1, max_batch=3, max_src_len=16, max_mel_len=96, lib_path=emu_lib)
    with pytest.raises(NotImplementedError):
        Trainer(im).imaml_step([([t1[0]], [t1[1]])], total_tasks=1)
