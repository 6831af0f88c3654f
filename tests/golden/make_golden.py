#!/usr/bin/env python3
"""Generate golden fixtures by importing the reference's own model code.

BUILD-CONTAINER ONLY: needs /root/reference (absent on the GPU box).  The reference has no
tests or golden vectors of its own, so the fixtures are outputs of the
reference's ``FastSpeech2`` / ``FastSpeech2Loss`` / ``get_scheduler`` / ``get_optimizer`` run
here on the deterministic synthetic inputs of ``meta_tts_amd.synth``.  Only data is stored
(inputs are regenerated from seeds, outputs are saved); no reference source is copied.

Import recipe = SURVEY.md Appendix B: four ``sys.modules`` shims for packages the image lacks
(unidecode, inflect, pytorch_lightning, resemblyzer) and a temp ``preprocessed_path`` holding
stats.json / speakers.json.  The MAML loop is restated over ``torch.func.functional_call`` on
the reference model because learn2learn is not installed (parity unpinned at that rule only).

Usage:  python tests/golden/make_golden.py            (writes tests/golden/*.npz)
"""
import json
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)

from meta_tts_amd import synth  # noqa: E402
from meta_tts_amd.config import (ModelDims, SYNTH_N_SPEAKER, SYNTH_STATS, load_yaml)  # noqa: E402


def install_shims():
    m = types.ModuleType("unidecode"); m.unidecode = lambda s: s; sys.modules["unidecode"] = m
    m = types.ModuleType("inflect"); m.engine = lambda: None; sys.modules["inflect"] = m
    pl = types.ModuleType("pytorch_lightning")

    class LightningModule(torch.nn.Module):
        def freeze(self):
            for p in self.parameters():
                p.requires_grad = False
            self.eval()

    pl.LightningModule = LightningModule
    sys.modules["pytorch_lightning"] = pl
    m = types.ModuleType("resemblyzer")

    class VoiceEncoder(torch.nn.Module):
        def __init__(self, *a, **k):
            super().__init__()

    m.VoiceEncoder = VoiceEncoder
    sys.modules["resemblyzer"] = m


def build_reference_model(pitch_level="phoneme_level", energy_level="phoneme_level"):
    install_shims()
    if REF not in sys.path:
        sys.path.insert(0, REF)
    pre = load_yaml(os.path.join(REF, "config/preprocess/LibriTTS.yaml"))
    pre["preprocessing"]["pitch"]["feature"] = pitch_level
    pre["preprocessing"]["energy"]["feature"] = energy_level
    mod = load_yaml(os.path.join(REF, "config/model/base.yaml"))
    alg = load_yaml(os.path.join(REF, "config/algorithm/meta_emb_vad.yaml"))
    trn = load_yaml(os.path.join(REF, "config/train/base.yaml"))
    tmp = tempfile.mkdtemp(prefix="mtts_golden_")
    with open(os.path.join(tmp, "stats.json"), "w") as f:
        json.dump(SYNTH_STATS, f)
    with open(os.path.join(tmp, "speakers.json"), "w") as f:
        json.dump({str(i): i for i in range(SYNTH_N_SPEAKER)}, f)
    pre["path"]["preprocessed_path"] = tmp
    from lightning.model.fastspeech2 import FastSpeech2
    from lightning.model.loss import FastSpeech2Loss
    model = FastSpeech2(pre, mod, alg)
    loss_fn = FastSpeech2Loss(pre, mod)
    pre_ph = load_yaml(os.path.join(REF, "config/preprocess/LibriTTS.yaml"))   # ModelDims only reads sizes from it
    dims = ModelDims(mod, pre_ph)
    params = synth.make_params(dims, seed=0)
    sd = model.state_dict()
    # how far numpy linspace (our bins) is from torch.linspace (reference bins)
    bins_dev = float(np.abs(sd["variance_adaptor.pitch_bins"].numpy() - params["variance_adaptor.pitch_bins"]).max())
    pos_dev = float(np.abs(sd["encoder.position_enc"].numpy() - params["encoder.position_enc"]).max())
    missing = [k for k in params if k not in sd]
    extra = [k for k in sd if k not in params and "running" not in k and "num_batches" not in k]
    assert not missing and not extra, (missing, extra)
    with torch.no_grad():
        for k, v in params.items():
            assert tuple(sd[k].shape) == v.shape, (k, sd[k].shape, v.shape)
            sd[k].copy_(torch.from_numpy(v))
    model.load_state_dict(sd)
    return model, loss_fn, dims, (pre, mod, alg, trn), {"bins_dev": bins_dev, "pos_dev": pos_dev}


def patch_dropout_identity(model):
    """nn.Dropout.p = 0 and F.dropout -> identity (PostNet hard-codes it)."""
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    import transformer.Layers as L
    L.F = types.SimpleNamespace(**{k: getattr(torch.nn.functional, k) for k in dir(torch.nn.functional)})
    L.F.dropout = lambda x, p=0.5, training=True, inplace=False: x


def tb(batch):
    out = []
    for i, x in enumerate(batch):
        if isinstance(x, np.ndarray):
            out.append(torch.from_numpy(x))
        else:
            out.append(int(x) if i in (5, 8) else x)
    return tuple(out)


def reset_bn(model):
    for m in model.modules():
        if isinstance(m, torch.nn.BatchNorm1d):
            m.reset_running_stats()


def grads_of(model, loss):
    names = [n for n, p in model.named_parameters() if p.requires_grad]
    ps = [p for n, p in model.named_parameters() if p.requires_grad]
    gs = torch.autograd.grad(loss, ps, allow_unused=True)
    return {n: (g if g is not None else torch.zeros_like(p)) for n, g, p in zip(names, gs, ps)}


FULL_GRADS = ["mel_linear.weight", "decoder.layer_stack.5.pos_ffn.w_2.weight",
              "variance_adaptor.duration_predictor.linear_layer.weight",
              "encoder.layer_stack.0.slf_attn.w_qs.bias", "postnet.convolutions.4.1.weight"]

def head(g):
    """Keep fixtures small: the first 4 rows of a >=2-D gradient, all of a vector."""
    g = g.detach().numpy()
    return g[:4].copy() if g.ndim >= 2 else g.copy()


SMALL = dict(s_range=(6, 13), d_range=(1, 7), first_len=12)

def load_synth_params(model, dims, weight_scale=1.0, edit=None):
    params = synth.make_params(dims, seed=0, weight_scale=weight_scale)
    if edit:
        edit(params)
    sd = model.state_dict()
    with torch.no_grad():
        for k, v in params.items():
            sd[k].copy_(torch.from_numpy(v))
    model.load_state_dict(sd)


def maml_fixture(model, loss_fn, modules, lr):
    """5 inner steps on the SMALL task with the reference model, first- and second-order (MAML rule restated over
    functional_call: learn2learn is not installed)."""
    from torch.func import functional_call
    res = {}
    for order in ("fo", "so"):
        model.train(); reset_bn(model)
        sup = tb(synth.make_batch(21, 3, speaker=9, **SMALL))
        qry = tb(synth.make_batch(22, 3, speaker=9, **SMALL))
        named = dict(model.named_parameters())
        frozen = ("position_enc", "pitch_bins", "energy_bins")
        names = [k for k in named if k.split(".")[0] in modules and not k.endswith(frozen)]
        fast = {k: named[k] for k in names}
        sup_losses = []
        for _ in range(5):
            preds = functional_call(model, fast, sup[2:])
            l = loss_fn(sup, preds)
            sup_losses.append([float(x) for x in l])
            gr = torch.autograd.grad(l[0], [fast[k] for k in names], create_graph=(order == "so"))
            fast = {k: fast[k] - lr * g_ for k, g_ in zip(names, gr)}
        # query pass: support speaker ids, mean speaker embedding (base_adaptor.py:66-67,122)
        spk_w = fast["speaker_emb.model.weight"]
        mean_row = spk_w[sup[2]].mean(dim=0, keepdim=True)
        # the table path with a single-speaker task: mean of identical rows == that row, so the
        # reference module can be called with the support ids directly (same batch size)
        preds = functional_call(model, fast, (sup[2],) + qry[3:])
        ql = loss_fn(qry, preds)
        ps = [p for n, p in model.named_parameters() if p.requires_grad]
        pn = [n for n, p in model.named_parameters() if p.requires_grad]
        og = torch.autograd.grad(ql[0], ps, allow_unused=True)
        og = {n: (g_ if g_ is not None else torch.zeros_like(p)) for n, g_, p in zip(pn, og, ps)}
        res[f"{order}_sup_losses"] = np.array(sup_losses, np.float64)
        res[f"{order}_qry_losses"] = np.array([float(x) for x in ql], np.float64)
        res[f"{order}_delta_norms"] = np.array(
            [float((fast[k] - named[k]).detach().double().norm()) for k in names], np.float64)
        res[f"{order}_outer_names"] = np.array(pn)
        res[f"{order}_outer_norms"] = np.array([float(v.double().norm()) for v in og.values()], np.float64)
        for n in FULL_GRADS:
            res[f"{order}_grad::" + n] = head(og[n])
        res[f"{order}_grad::speaker_row"] = og["speaker_emb.model.weight"][9].numpy()
        res[f"{order}_qry_mel_post"] = preds[1].detach().numpy()
        res[f"{order}_mean_row_check"] = (mean_row - spk_w[9]).abs().max().detach().numpy()
    res["adapted_names"] = np.array(names)
    return res


def frame_level_fixture():
    """Frame-level pitch / energy (preprocess `feature: frame_level`; modules.py:139-148, loss.py:54-63): the reference model built
    with that preprocess config on the padded SMALL batch with one value per mel frame — losses, predictions, every gradient
    norm (train mode), and the loss 6-tuple of the two mixed configurations."""
    out = {}
    for tag, pl, el in (("ff", "frame_level", "frame_level"), ("pf", "phoneme_level", "frame_level"), ("fp", "frame_level", "phoneme_level")):
        model, loss_fn, dims, cfgs, _ = build_reference_model(pl, el)
        patch_dropout_identity(model)
        batch = synth.make_batch(11, 3, speaker=5, pitch_level=pl, energy_level=el, **SMALL)
        b = tb(batch)
        model.train(); reset_bn(model)
        o = model(*b[2:])
        lo = loss_fn(b, o)
        out[f"{tag}_losses"] = np.array([float(x) for x in lo], np.float64)
        out[f"{tag}_mel_post"] = o[1].detach().numpy()
        out[f"{tag}_p"] = o[2].detach().numpy(); out[f"{tag}_e"] = o[3].detach().numpy()
        if tag == "ff":
            g = grads_of(model, lo[0])
            out["ff_grad_names"] = np.array(list(g.keys()))
            out["ff_grad_norms"] = np.array([float(v.double().norm()) for v in g.values()], np.float64)
            for n in FULL_GRADS:
                out["ff_grad::" + n] = head(g[n])
            model.eval()
            with torch.no_grad():
                oe = model(*b[2:])
                fr = model(*b[2:6], p_control=1.1, e_control=0.9)
            out["ff_eval_mel_post"] = oe[1].numpy()
            out["ff_fr_mel_post"] = fr[1].numpy(); out["ff_fr_d_rounded"] = fr[5].numpy(); out["ff_fr_mel_len"] = fr[9].numpy()
            out["ff_fr_p"] = fr[2].numpy(); out["ff_fr_e"] = fr[3].numpy()
    return out


def c5_edit(params):
    """The random-init duration predictor emits ~0 frames; bias ln(8) and a damped weight give LibriTTS-like durations
    (about 7 frames per phoneme) — the same edit bench.py's inference leg applies."""
    params["variance_adaptor.duration_predictor.linear_layer.bias"][:] = np.log(8.0)
    params["variance_adaptor.duration_predictor.linear_layer.weight"] *= 0.25


def c5_fixture(model, dims):
    """BASELINE config 5, synthesis half: free-running forward (modules.py:132-137,150-158) of the C1 utterance and of a
    padded 3-utterance batch with the duration predictor of c5_edit, eval mode and train mode (the reference synthesises
    with the adapted clone left in .train(), base_adaptor.py:170-189), with p/e/d controls != 1."""
    load_synth_params(model, dims, edit=c5_edit)
    out = {}
    cases = (("c1", synth.make_batch(0, 1), (1.0, 1.0, 1.0)), ("b3", synth.make_batch(7, 3, speaker=11), (1.2, 0.8, 0.9)))
    for tag, batch, (pc, ec, dc) in cases:
        b = tb(batch)
        for mode in ("eval", "train"):
            model.train(mode == "train"); reset_bn(model)
            with torch.no_grad():
                fr = model(*b[2:6], p_control=pc, e_control=ec, d_control=dc)
            k = f"{tag}_{mode}_"
            out[k + "mel_post"] = fr[1].numpy(); out[k + "mel"] = fr[0].numpy()
            out[k + "p"] = fr[2].numpy(); out[k + "e"] = fr[3].numpy(); out[k + "logd"] = fr[4].numpy()
            out[k + "d_rounded"] = fr[5].numpy(); out[k + "mel_len"] = fr[9].numpy()
        out[tag + "_controls"] = np.array([pc, ec, dc], np.float64)
    load_synth_params(model, dims)
    return out



def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    model, loss_fn, dims, cfgs, devs = build_reference_model()
    patch_dropout_identity(model)
    out_dir = HERE
    meta = {"torch": torch.__version__, **devs}

    # ---------------- C1: single utterance, full size --------------------------------------
    batch = synth.make_batch(0, 1)
    b = tb(batch)
    model.eval(); reset_bn(model)
    with torch.no_grad():
        o = model(*b[2:])
        lo = loss_fn(b, o)
        fr = model(*b[2:6])  # free-running: durations from the predictor
    c1 = {"mel": o[0].numpy(), "mel_post": o[1].numpy(), "p": o[2].numpy(), "e": o[3].numpy(),
          "logd": o[4].numpy(), "losses": np.array([float(x) for x in lo], np.float64),
          "fr_d_rounded": fr[5].numpy(), "fr_mel_len": fr[9].numpy(),
          "fr_mel_post": fr[1].numpy(), "fr_p": fr[2].numpy(), "fr_e": fr[3].numpy()}
    model.train(); reset_bn(model)
    with torch.no_grad():
        o = model(*b[2:])
        lo = loss_fn(b, o)
    c1.update({"train_mel_post": o[1].numpy(), "train_losses": np.array([float(x) for x in lo], np.float64)})
    if not (os.environ.get("MTTS_GOLDEN_ONLY_MAML") or os.environ.get("MTTS_GOLDEN_ONLY_C5")):
        np.savez_compressed(os.path.join(out_dir, "c1_forward.npz"), **c1)
    meta["c1_shapes"] = {"S": int(batch[5]), "T": int(batch[8])}

    # ---------------- small padded batch: forward, loss, gradients (train mode) ---------------
    batch = synth.make_batch(11, 3, speaker=5, **SMALL)
    # bucketize edge cases: at / below min, at / above max
    batch[9][0, :4] = np.array([dims.pitch_min, dims.pitch_min - 1.0, dims.pitch_max, dims.pitch_max + 1.0], np.float32)
    batch[10][0, :4] = np.array([dims.energy_min, dims.energy_min - 1.0, dims.energy_max, dims.energy_max + 1.0], np.float32)
    b = tb(batch)
    model.train(); reset_bn(model)
    o = model(*b[2:])
    lo = loss_fn(b, o)
    g = grads_of(model, lo[0])
    small = {"mel": o[0].detach().numpy(), "mel_post": o[1].detach().numpy(), "p": o[2].detach().numpy(),
             "e": o[3].detach().numpy(), "logd": o[4].detach().numpy(),
             "losses": np.array([float(x) for x in lo], np.float64),
             "grad_names": np.array(list(g.keys())),
             "grad_norms": np.array([float(v.double().norm()) for v in g.values()], np.float64),
             "p_targets": batch[9], "e_targets": batch[10]}
    for n in FULL_GRADS:
        small["grad::" + n] = head(g[n])
    small["grad::speaker_row"] = g["speaker_emb.model.weight"][5].numpy()
    small["grad::src_word_emb_rows"] = g["encoder.src_word_emb.weight"][:8].numpy()
    sd = model.state_dict()
    for i in range(5):
        small[f"bn{i}_running_mean"] = sd[f"postnet.convolutions.{i}.1.running_mean"].numpy().copy()
        small[f"bn{i}_running_var"] = sd[f"postnet.convolutions.{i}.1.running_var"].numpy().copy()
    # eval-mode forward of the same batch (BN running stats just updated once)
    model.eval()
    with torch.no_grad():
        oe = model(*b[2:])
    small["eval_mel_post"] = oe[1].numpy()
    if not (os.environ.get("MTTS_GOLDEN_ONLY_MAML") or os.environ.get("MTTS_GOLDEN_ONLY_C5")):
        np.savez_compressed(os.path.join(out_dir, "small_grad.npz"), **small)

    # ---------------- frame-level pitch / energy ------------------------------------------------------
    if os.environ.get("MTTS_GOLDEN_ONLY_FRAME"):
        np.savez_compressed(os.path.join(out_dir, "frame_level.npz"), **frame_level_fixture())
        return

    # ---------------- C5: free-running synthesis with realistic predicted durations --------------
    if os.environ.get("MTTS_GOLDEN_ONLY_C5") or not os.environ.get("MTTS_GOLDEN_ONLY_MAML"):
        np.savez_compressed(os.path.join(out_dir, "c5_synth.npz"), **c5_fixture(model, dims))
        if os.environ.get("MTTS_GOLDEN_ONLY_C5"):
            return

    # ---------------- MAML: 5 inner steps, FO and SO, small task ------------------------------
    alg = cfgs[2]
    modules = alg["adapt"]["modules"]
    # lr 1e-4: contractive inner loop (losses fall) -> tight parity; 1e-3 / 2e-3: the tiny random model is expansive
    # there (support loss grows over the steps), kept as loose-tolerance cases; "lr1e-3_scaled": the reference's own inner
    # lr on weights scaled by 0.5 (synth.make_params(weight_scale=0.5)), where the five steps are contractive
    # (7.5 -> 3.8), so the reference configuration is pinned tightly too
    only = os.environ.get("MTTS_GOLDEN_ONLY_MAML")  # e.g. "lr1e-4": write only that fixture (the others stay as committed)
    for tag, lr, wscale in (("lr1e-4", 0.0001, 1.0), ("lr1e-3", 0.001, 1.0), ("lr2e-3", 0.002, 1.0), ("lr1e-3_scaled", 0.001, 0.5)):
        if only and tag != only:
            continue
        load_synth_params(model, dims, weight_scale=wscale)
        res = maml_fixture(model, loss_fn, modules, lr)
        np.savez_compressed(os.path.join(out_dir, f"maml_small_{tag}.npz"), **res)
    load_synth_params(model, dims)

    if os.environ.get("MTTS_GOLDEN_ONLY_MAML"):
        return
    # ---------------- optimizer / scheduler ---------------------------------------------------
    from lightning.scheduler import get_scheduler
    from lightning.optimizer import get_optimizer
    trn, mod = cfgs[3], cfgs[1]
    lin = torch.nn.Linear(4, 3)
    with torch.no_grad():
        lin.weight.copy_(torch.arange(12, dtype=torch.float32).view(3, 4) * 0.1 - 0.5)
        lin.bias.copy_(torch.tensor([0.1, -0.2, 0.3]))
    opt = get_optimizer(lin, mod, trn)
    sch = get_scheduler(opt, trn)
    lam = sch.lr_lambdas[0]
    steps = [0, 1, 3998, 3999, 4000, 299999, 300000, 300001, 400000, 500001]
    lrs = [float(opt.defaults["lr"] * lam(s)) for s in steps]
    traj = []
    g = np.random.RandomState(5)
    grads_w = []
    for it in range(3):
        gw = torch.from_numpy(g.standard_normal((3, 4)).astype(np.float32)) * (10.0 if it == 1 else 0.1)
        gb = torch.from_numpy(g.standard_normal((3,)).astype(np.float32)) * 0.1
        lin.weight.grad = gw.clone(); lin.bias.grad = gb.clone()
        norm = torch.nn.utils.clip_grad_norm_(lin.parameters(), trn["optimizer"]["grad_clip_thresh"])
        opt.step(); sch.step()
        grads_w.append(np.concatenate([gw.numpy().ravel(), gb.numpy().ravel()]))
        traj.append(np.concatenate([lin.weight.detach().numpy().ravel(), lin.bias.detach().numpy().ravel(),
                                    [float(norm)], [float(sch.get_last_lr()[0])]]))
    np.savez_compressed(os.path.join(out_dir, "optimizer.npz"), steps=np.array(steps), lrs=np.array(lrs, np.float64),
                        grads=np.array(grads_w), traj=np.array(traj, np.float64),
                        init=np.concatenate([(np.arange(12) * 0.1 - 0.5).astype(np.float32), [0.1, -0.2, 0.3]]))

    with open(os.path.join(out_dir, "meta.json"), "w") as f:
        json.dump(meta, f, indent=1)
    print("golden fixtures written to", out_dir, meta)


if __name__ == "__main__":
    main()
