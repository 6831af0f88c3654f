"""d-vector speaker encoder host layer — mirror of `SpeakerEncoder` for `speaker_emb: dvec`
(reference lightning/model/speaker_encoder.py:33-76, config/algorithm/dvec.yaml).

In the reference the `dvec` mode owns a frozen resemblyzer `VoiceEncoder('cpu')` (LSTM(40, 256, 3) + Linear(256, 256) + ReLU, the
architecture its own GE2E class restates, :11-31) and is called with `speaker_args = (ref_mels, ref_slices)` — the
concatenated 160-frame partial utterances of the batch and one python slice per utterance (lightning/collate.py:29-43).  This
class keeps that call signature, runs the encoder in libmtts.so (include/mtts.h, mtts_dvector_*) and returns the (B, 256)
embeddings the acoustic model adds to the encoder output (fastspeech2.py:65-68,91-94); they reach the engine through
`mtts_batch.spk_emb`.  The trained variants (`encoder`, `scratch_encoder`) need the LSTM backward and raise."""
from __future__ import annotations

import ctypes as C
import zlib

import numpy as np

from . import _lib
from .engine import MttsError

MEL_N_CHANNELS, HIDDEN, EMBED, LAYERS, PARTIAL_FRAMES = 40, 256, 256, 3, 160   # speaker_encoder.py:11-14; resemblyzer partials_n_frames


def tensor_shapes(n_mels=MEL_N_CHANNELS, hidden=HIDDEN, emb=EMBED, layers=LAYERS):
    """{state-dict name: shape} in torch's nn.LSTM / nn.Linear naming."""
    out = {}
    for k in range(layers):
        out[f"lstm.weight_ih_l{k}"] = (4 * hidden, n_mels if k == 0 else hidden)
        out[f"lstm.weight_hh_l{k}"] = (4 * hidden, hidden)
        out[f"lstm.bias_ih_l{k}"] = (4 * hidden,)
        out[f"lstm.bias_hh_l{k}"] = (4 * hidden,)
    out["linear.weight"] = (emb, hidden)
    out["linear.bias"] = (emb,)
    return out


def synthetic_state_dict(seed=0, **kw):
    """Deterministic weights with torch's default LSTM / Linear initialisation scale (uniform +-1/sqrt(hidden))."""
    sd = {}
    hidden = kw.get("hidden", HIDDEN)
    for name, shape in tensor_shapes(**kw).items():
        g = np.random.RandomState((zlib.crc32(name.encode()) ^ seed) & 0x7FFFFFFF)
        sd[name] = ((2 * g.rand(*shape) - 1) / np.sqrt(hidden)).astype(np.float32)
    return sd


class DVectorEncoder:
    """`SpeakerEncoder` with emb_type == "dvec": __call__((ref_mels, ref_slices)) -> (B, emb) float32."""

    def __init__(self, state_dict=None, max_partials=256, max_utts=64, n_mels=MEL_N_CHANNELS, hidden=HIDDEN, emb=EMBED, layers=LAYERS,
                 frames=PARTIAL_FRAMES, device=0, lib_path=None):
        self.lib = _lib.load(lib_path)
        self.cfg = dict(n_mels=n_mels, hidden=hidden, emb=emb, layers=layers)
        self.frames, self.max_partials, self.max_utts = frames, max_partials, max_utts
        h = C.c_void_p()
        if self.lib.mtts_dvector_create(n_mels, hidden, layers, emb, max_partials, frames, max_utts, device, C.byref(h)) != 0:
            raise MttsError(self.lib.mtts_dvector_last_error(None).decode())
        self.h = h
        self.load_state_dict(state_dict if state_dict is not None else synthetic_state_dict(**self.cfg))

    def set_stream(self, stream_ptr: int):
        """HIP stream of this encoder's launches — the engine's, when the encoder is trained with it (shared device scalars)."""
        self._check(self.lib.mtts_dvector_set_stream(self.h, C.c_void_p(stream_ptr)))

    def close(self):
        if getattr(self, "h", None):
            self.lib.mtts_dvector_destroy(self.h)
            self.h = None

    __del__ = close

    def _check(self, rc):
        if rc != 0:
            raise MttsError(self.lib.mtts_dvector_last_error(self.h).decode())

    def load_state_dict(self, sd, prefix=""):
        """Accepts the encoder's own names or the reference checkpoint's (`model.speaker_emb.model.lstm.weight_ih_l0`, ...)."""
        for name, shape in tensor_shapes(**self.cfg).items():
            key = next((k for k in (prefix + name, "speaker_emb.model." + name, "model.speaker_emb.model." + name) if k in sd), None)
            if key is None:
                raise KeyError(f"d-vector tensor {name} missing from the state dict")
            a = np.ascontiguousarray(np.asarray(sd[key], np.float32))
            if a.shape != shape:
                raise ValueError(f"{key}: shape {a.shape}, expected {shape}")
            self._check(self.lib.mtts_dvector_load(self.h, name.encode(), a.ctypes.data_as(C.c_void_p), a.size))
        self.state = {n: np.asarray(sd[next(k for k in (prefix + n, "speaker_emb.model." + n, "model.speaker_emb.model." + n) if k in sd)], np.float32)
                      for n in tensor_shapes(**self.cfg)}

    def state_dict(self):
        return dict(self.state)

    def embed(self, ref_mels, ref_slices, return_partials=False):
        ref_mels = np.ascontiguousarray(np.asarray(ref_mels, np.float32))
        if ref_mels.ndim != 3 or ref_mels.shape[1] != self.frames or ref_mels.shape[2] != self.cfg["n_mels"]:
            raise ValueError(f"ref_mels must be (n_partials, {self.frames}, {self.cfg['n_mels']}), got {ref_mels.shape}")
        n = ref_mels.shape[0]
        off = [0]
        for sl in ref_slices:   # the collate builds consecutive slices (collate.py:33-38)
            start, stop = (sl.start or 0), sl.stop
            if start != off[-1] or stop < start:
                raise ValueError("ref_slices must be consecutive, non-overlapping slices starting at 0")
            off.append(stop)
        if off[-1] != n:
            raise ValueError("ref_slices must cover every partial utterance")
        if n > self.max_partials or len(ref_slices) > self.max_utts:
            raise ValueError("batch exceeds the encoder's capacity (max_partials / max_utts)")
        off = np.asarray(off, np.int32)
        out = np.empty((len(ref_slices), self.cfg["emb"]), np.float32)
        part = np.empty((n, self.cfg["emb"]), np.float32) if return_partials else None
        self._check(self.lib.mtts_dvector_embed(self.h, ref_mels.ctypes.data_as(C.c_void_p), n, off.ctypes.data_as(C.c_void_p), len(ref_slices),
                                                out.ctypes.data_as(C.c_void_p), part.ctypes.data_as(C.c_void_p) if return_partials else None))
        return (out, part) if return_partials else out

    def __call__(self, args):
        ref_mels, ref_slices = args
        return self.embed_train(ref_mels, ref_slices) if self.training else self.embed(ref_mels, ref_slices)

    # ---- trained variants: speaker_emb: encoder / scratch_encoder (speaker_encoder.py:54-55,59-60) -------------------------------
    training = False   # True: __call__ keeps the state the backward sweep needs

    def enable_training(self):
        self._check(self.lib.mtts_dvector_enable_training(self.h))
        self.training = True

    def _offsets(self, ref_mels, ref_slices):
        ref_mels = np.ascontiguousarray(np.asarray(ref_mels.detach().cpu().numpy() if hasattr(ref_mels, "detach") else ref_mels, np.float32))
        off = np.asarray([0] + [sl.stop for sl in ref_slices], np.int32)
        assert all((sl.start or 0) == off[i] for i, sl in enumerate(ref_slices)) and off[-1] == ref_mels.shape[0]
        return ref_mels, off

    def embed_train(self, ref_mels, ref_slices):
        ref_mels, off = self._offsets(ref_mels, ref_slices)
        out = np.empty((len(ref_slices), self.cfg["emb"]), np.float32)
        self._check(self.lib.mtts_dvector_embed_train(self.h, ref_mels.ctypes.data_as(C.c_void_p), ref_mels.shape[0], off.ctypes.data_as(C.c_void_p),
                                                      len(ref_slices), out.ctypes.data_as(C.c_void_p)))
        return out

    def backward(self, dout):
        """dout (B, emb): dLoss/d(embeddings of the last embed_train) -> parameter gradients (`export(name, 1)`)."""
        d = np.ascontiguousarray(np.asarray(dout, np.float32))
        self._check(self.lib.mtts_dvector_backward(self.h, d.ctypes.data_as(C.c_void_p)))

    def export(self, name, which=0):
        out = np.empty(tensor_shapes(**self.cfg)[name], np.float32)
        self._check(self.lib.mtts_dvector_export(self.h, name.encode(), which, out.ctypes.data_as(C.c_void_p), out.size))
        return out

    def import_state(self, name, which, value):
        a = np.ascontiguousarray(np.asarray(value, np.float32))
        self._check(self.lib.mtts_dvector_import(self.h, name.encode(), which, a.ctypes.data_as(C.c_void_p), a.size))
        if which == 0:
            self.state[name] = a.reshape(tensor_shapes(**self.cfg)[name]).copy()

    def grad_sumsq_ptr(self):
        """Device address of sum(grad^2) — the encoder's term of the joint clip_grad_norm_ (main.py:61)."""
        return self.lib.mtts_dvector_grad_sumsq(self.h)

    def adam_step(self, norm_dev, max_norm, lr, betas=(0.9, 0.98), eps=1e-9, weight_decay=0.0):
        self._check(self.lib.mtts_dvector_adam_step(self.h, norm_dev, float(max_norm), float(lr), float(betas[0]), float(betas[1]), float(eps),
                                                    float(weight_decay)))
        self.state = {n: self.export(n, 0) for n in tensor_shapes(**self.cfg)}

    def set_optimizer_step(self, step):
        self._check(self.lib.mtts_dvector_set_optimizer_step(self.h, int(step)))
