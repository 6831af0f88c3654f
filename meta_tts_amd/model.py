"""Reference-shaped Python face of the hot path (the drop-in boundary of SURVEY.md section 8(b)).

Class names, constructor / forward signatures, the 12-tuple batch and 10-tuple prediction layouts and
the state_dict key names are the reference's (lightning/model/fastspeech2.py:16-112,
lightning/model/loss.py:5-92); every number is produced by libmtts on the MI355X — these classes
hold no arithmetic of their own and raise if the library or a GPU is missing.
"""
from __future__ import annotations

import json
import os
from typing import Dict, Optional, Sequence

import numpy as np

from . import synth
from .config import ModelDims, SYNTH_N_SPEAKER, SYNTH_STATS
from .engine import LOSS_NAMES, Engine, MttsError


class Predictions(tuple):
    """The reference's 10-tuple (fastspeech2.py:101-112) plus the handle that produced it, so that
    FastSpeech2Loss can evaluate the loss where the data lives (device) instead of re-reading it."""
    engine: Engine
    slot: int
    task: int
    epoch: int  # Engine.epoch when these predictions were produced; the loss refuses to run once the engine has moved on


def _read_preprocessed(preprocess_config) -> tuple:
    """stats.json / speakers.json under preprocessed_path (modules.py:41-46, speaker_encoder.py:49-50);
    the synthetic defaults are used when the directory does not exist (no corpus in this image)."""
    path = preprocess_config.get("path", {}).get("preprocessed_path", "")
    stats, n_spk = SYNTH_STATS, SYNTH_N_SPEAKER
    sp = os.path.join(path, "stats.json")
    if os.path.exists(sp):
        with open(sp) as f:
            stats = json.load(f)
    kp = os.path.join(path, "speakers.json")
    if os.path.exists(kp):
        with open(kp) as f:
            n_spk = len(json.load(f))
    return stats, n_spk


class FastSpeech2:
    """lightning/model/fastspeech2.py:16 — engine-backed.  ``forward`` is teacher-forced when duration targets
    and mels are given, free-running otherwise (predicted durations size the output)."""

    def __init__(self, preprocess_config, model_config, algorithm_config, *, max_tasks: int = 1, max_batch: int = 16,
                 max_src_len: int = 128, max_mel_len: Optional[int] = None, device: int = 0, lib_path: Optional[str] = None):
        spk_mode = algorithm_config["adapt"]["speaker_emb"]
        if spk_mode not in ("table", "shared", "dvec", "encoder", "scratch_encoder"):
            raise MttsError("adapt.speaker_emb must be one of table / shared / dvec / encoder / scratch_encoder (speaker_encoder.py:45-60)")
        if spk_mode in ("encoder", "scratch_encoder") and algorithm_config.get("type", "baseline") != "baseline":
            raise MttsError("a trained speaker encoder (speaker_emb: encoder / scratch_encoder) is only supported by the baseline system, as in "
                            "config/algorithm/{encoder,scratch_encoder}.yaml: the MAML passes have no tangent / inner-loop path through the LSTM")
        if algorithm_config["adapt"]["type"] != "spk":
            raise MttsError("adapt.type == 'lang' (codebook phoneme embedding) is out of scope (SURVEY.md #8)")
        stats, n_spk = _read_preprocessed(preprocess_config)
        if spk_mode in ("shared", "dvec", "encoder", "scratch_encoder"):
            n_spk = 1   # shared: nn.Embedding(1, d), one vector for every speaker (speaker_encoder.py:52-53); dvec: no table at all (the
                        # engine's one-row table is an unused placeholder, the embeddings come with the batch)
        self.spk_mode = spk_mode
        self.dims = ModelDims(model_config, preprocess_config, n_speaker=n_spk, stats=stats)
        self.model_config, self.preprocess_config, self.algorithm_config = model_config, preprocess_config, algorithm_config
        self.adapt_modules = tuple(algorithm_config["adapt"].get("modules", ()))
        self.engine = Engine(self.dims, adapt_modules=self.adapt_modules, max_tasks=max_tasks, max_B=max_batch,
                             max_S=max_src_len, max_T=max_mel_len or self.dims.max_seq_len, device=device, lib_path=lib_path,
                             shared_speaker=(spk_mode == "shared"))
        self.speaker_encoder = None
        if spk_mode in ("dvec", "encoder", "scratch_encoder"):   # LSTM speaker encoder in front of the acoustic model (speaker_encoder.py:54-60,
            # 71-76): frozen for dvec (`self.freeze()`), trained with the model for encoder (pretrained init) / scratch_encoder
            from .speaker_encoder import DVectorEncoder
            # the reference's encoder is fixed at 40 mels / 3 x 256 / 160-frame partials and emits encoder_hidden = 256 values; the optional
            # `adapt.dvector` block (not a reference key) only exists so that tests can run a small one
            kw = dict(algorithm_config["adapt"].get("dvector", {}))
            self.speaker_encoder = DVectorEncoder(max_partials=max(256, 16 * max_batch), max_utts=max_batch, emb=self.dims.d_model, device=device,
                                                  lib_path=lib_path, **kw)
            self.engine.speaker_encoder = self.speaker_encoder
            if spk_mode != "dvec":
                self.speaker_encoder.enable_training()
        self.training = True
        self.load_state_dict(synth.make_params(self.dims, seed=0), strict=self.speaker_encoder is None)

    # -- nn.Module-like surface ---------------------------------------------------------
    def train(self, mode: bool = True):
        self.training = mode
        if self.speaker_encoder is not None and self.spk_mode != "dvec":
            self.speaker_encoder.training = bool(mode)   # keep the BPTT state only for training forwards
        return self

    def eval(self):
        return self.train(False)

    def state_dict(self) -> Dict[str, np.ndarray]:
        sd = dict(self.engine.state_dict())
        pos = synth.sinusoid_table(self.dims.max_seq_len + 1, self.dims.d_model)[None]
        sd["encoder.position_enc"], sd["decoder.position_enc"] = pos, pos.copy()
        sd["variance_adaptor.pitch_bins"], sd["variance_adaptor.energy_bins"] = self.engine.get_bins()  # what the engine really quantises with
        for i in range(self.dims.postnet_layers):
            m, v, t = self.engine.get_bn_buffers(i)
            sd[f"postnet.convolutions.{i}.1.running_mean"] = m
            sd[f"postnet.convolutions.{i}.1.running_var"] = v
            sd[f"postnet.convolutions.{i}.1.num_batches_tracked"] = np.array(t, np.int64)
        if self.speaker_encoder is not None:   # the reference's keys: speaker_emb.model.{lstm,linear}.* instead of an embedding table
            sd.pop("speaker_emb.model.weight", None)
            sd.update({"speaker_emb.model." + k: v for k, v in self.speaker_encoder.state_dict().items()})
        return sd

    def load_state_dict(self, sd: Dict[str, np.ndarray], strict: bool = True):
        sd = {k: (v.detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v)) for k, v in sd.items()}
        if self.speaker_encoder is not None:
            sd = dict(sd)
            sd["speaker_emb.model.weight"] = np.zeros((1, self.dims.d_model), np.float32)   # the engine's unused placeholder row
            if any(k.startswith("speaker_emb.model.lstm.") for k in sd):
                self.speaker_encoder.load_state_dict(sd, prefix="speaker_emb.model.")
            elif strict:
                raise KeyError("speaker_emb.model.lstm.* missing from the state dict (speaker_emb: dvec)")
        self.engine.load_params(sd, strict=strict)
        # frozen nn.Parameters that torch restores from the checkpoint (modules.py:57-71): a checkpoint trained on another
        # corpus carries that corpus' pitch / energy quantisation (system.py corpus-mismatch branch)
        pb, eb = sd.get("variance_adaptor.pitch_bins"), sd.get("variance_adaptor.energy_bins")
        if pb is not None or eb is not None:
            self.engine.set_bins(pb, eb)
        for i in range(self.dims.postnet_layers):
            km, kv, kt = (f"postnet.convolutions.{i}.1.{s}" for s in ("running_mean", "running_var", "num_batches_tracked"))
            if km in sd and kv in sd:
                self.engine.set_bn_buffers(i, sd[km], sd[kv], int(sd.get(kt, 0)))

    # -- forward ------------------------------------------------------------------------
    def forward(self, speaker_args, texts, src_lens, max_src_len, mels=None, mel_lens=None, max_mel_len=None,
                p_targets=None, e_targets=None, d_targets=None, p_control=1.0, e_control=1.0, d_control=1.0,
                *, slot: int = 0, use_fast: bool = False, spk_from=None, average_spk_emb: bool = False):
        batch = (None, None, speaker_args, texts, src_lens, max_src_len, mels, mel_lens, max_mel_len, p_targets, e_targets, d_targets)
        self.engine.set_batches(slot, [batch], spk_from=[spk_from] if spk_from is not None else None, average_spk=average_spk_emb)
        if d_targets is None or mels is None:  # free-running (fastspeech2.py forward without targets; modules.py:132-137)
            self.engine.synthesize(slot, use_fast=use_fast, train=self.training, p_control=p_control, e_control=e_control,
                                   d_control=d_control)
        else:
            self.engine.forward(slot, use_fast=use_fast, train=self.training)
        preds = self._predictions(slot, 0, batch)
        if d_targets is None or mels is None:  # the reference returns prediction * control (modules.py:86,97)
            preds[2].mul_(p_control)
            preds[3].mul_(e_control)
        return preds

    __call__ = forward

    def _predictions(self, slot: int, task: int, batch) -> Predictions:
        import torch
        o = self.engine.outputs(slot, task)
        B, S = self.engine.batch_shapes[slot][task]
        T = o["mel"].shape[1]
        src_lens = torch.as_tensor(np.asarray(batch[4]), dtype=torch.int64)
        mel_lens = torch.from_numpy(o["mel_lens"])
        src_masks = torch.arange(S)[None, :] >= src_lens[:, None]
        mel_masks = torch.arange(T)[None, :] >= mel_lens[:, None]
        d_rounded = torch.as_tensor(np.asarray(batch[11])) if batch[11] is not None else torch.from_numpy(o["d_rounded"])
        p = Predictions((torch.from_numpy(o["mel"]), torch.from_numpy(o["mel_post"]), torch.from_numpy(o["p"]),
                         torch.from_numpy(o["e"]), torch.from_numpy(o["logd"]), d_rounded, src_masks, mel_masks, src_lens, mel_lens))
        p.engine, p.slot, p.task, p.epoch = self.engine, slot, task, self.engine.epoch
        return p


class FastSpeech2Loss:
    """lightning/model/loss.py:5 — (total, mel, postnet mel, pitch, energy, duration), evaluated by the
    device-side reduction over the predictions still resident in HBM."""

    def __init__(self, preprocess_config, model_config):
        for k in ("pitch", "energy"):   # loss.py:9-14: both levels exist; the engine evaluates the loss at the configured one
            assert preprocess_config["preprocessing"][k]["feature"] in ("phoneme_level", "frame_level")

    def forward(self, inputs, predictions):
        import torch
        if not isinstance(predictions, Predictions):
            raise MttsError("FastSpeech2Loss needs the Predictions object returned by FastSpeech2.forward")
        if predictions.engine.epoch != predictions.epoch:
            raise MttsError("stale Predictions: the engine has run another batch / forward since these were produced "
                            "(the loss is evaluated on the activations resident in HBM)")
        vals = predictions.engine.loss(predictions.slot)[predictions.task]
        return tuple(torch.tensor(float(v)) for v in vals)

    __call__ = forward


def loss2dict(loss) -> Dict[str, float]:
    """lightning/utils.py:65-74"""
    return {k: float(v) for k, v in zip(LOSS_NAMES, loss)}
