"""ORACLE — test infrastructure, not product code.

Numpy restatement of the engine's counter-based dropout masks, so that the oracle can run the configuration the bench TIMES
(train-mode dropout on, as the reference trains: transformer/SubLayers.py:54,90 — nn.Dropout behind the attention `fc` and behind
`w_2`; lightning/model/modules.py:223,235 — dropout_1 / dropout_2 of the VariancePredictor behind each LayerNorm;
transformer/Layers.py:133-134 — F.dropout(., 0.5) behind every PostNet layer) with EXACTLY the masks the HIP kernels draw.

The reference draws its masks from torch's global Philox stream; no implementation can reproduce that stream element by element
inside fused kernels, and nothing in the reference depends on which Bernoulli stream is used.  What has to be pinned is that the
masks are applied at the reference's sites, with the reference's scaling x * keep / (1 - p), in forward AND in every backward /
Hessian-vector pass.  The engine's masks are a pure function (meta_tts_amd/csrc/rowops.h: splitmix64 / drop4, engine.h:
next_drop_seed / drop_spec), restated here:

    plan seed   k-th train-mode forward since mtts_set_dropout(h, 1, base):  next_drop_seed (engine.h)
    site seed   plan_seed * 0x9E3779B1 + site * 0x85EBCA6B + 0xC2B2AE35      (uint32)
    keep bits   16-bit fields of splitmix64(((site_seed << 32) ^ (task << 24)) + (row * C + col) / 4), field (row * C + col) % 4,
                keep <=> field >= round(p * 65536)

`row` is the row of the element in the engine's row space (engine.h header / plan.h): P = phoneme rectangle
G + b * (S_max + G) + s, F = packed frames foff[b] + t, R = mel rectangle G + b * (Tcap + G) + t, with G = 4 guard rows.
Sites: encoder layer l: 2l (attention), 2l + 1 (FFN); decoder layer l: 64 + 2l, 65 + 2l; duration / pitch / energy predictor:
128 / 132 / 136 (+1 for the second LayerNorm); PostNet layer i: 192 + i.
"""
from __future__ import annotations

from typing import Optional, Sequence

import numpy as np
import torch

G = 4                       # guard rows (plan.h: kPlanG)
M32 = 0xFFFFFFFF
_U64 = np.uint64


def plan_seed(base: int, k: int) -> int:
    """engine.h: next_drop_seed — the seed of the k-th (1-based) train-mode forward after mtts_set_dropout(h, 1, base)."""
    sd = ((((base + 0x9E3779B9) & M32) * 0x85EBCA6B) & M32) ^ ((0x632BE5AB * k) & M32)
    sd ^= sd >> 15
    return sd if sd else 1


def site_seed(pseed: int, site: int) -> int:
    """engine.h: drop_spec."""
    return (pseed * 0x9E3779B1 + site * 0x85EBCA6B + 0xC2B2AE35) & M32


def splitmix64(x: np.ndarray) -> np.ndarray:
    """rowops.h: splitmix64 (uint64 wrap-around arithmetic)."""
    with np.errstate(over="ignore"):
        x = x + _U64(0x9E3779B97F4A7C15)
        x = (x ^ (x >> _U64(30))) * _U64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> _U64(27))) * _U64(0x94D049BB133111EB)
        return x ^ (x >> _U64(31))


def keep_mask(sseed: int, task: int, rows: np.ndarray, C: int, prob: float) -> np.ndarray:
    """rowops.h: drop4 / dropout_kernel — bool [len(rows)][C]."""
    thr = int(round(prob * 65536.0))
    base = _U64(((sseed << 32) ^ (task << 24)) & 0xFFFFFFFFFFFFFFFF)
    eid = rows.astype(np.uint64)[:, None] * _U64(C) + np.arange(C, dtype=np.uint64)[None, :]
    with np.errstate(over="ignore"):
        h = splitmix64(base + eid // _U64(4))
    field = (h >> ((eid % _U64(4)) * _U64(16))) & _U64(0xFFFF)
    return field >= _U64(thr)


class DropoutMasks:
    """The masks of ONE train-mode forward (and of every pass that replays it) of task `task` in its launch group.

    fs2_forward binds the geometry (`bind`), then asks for `apply(x, site, space, prob)` at each of the reference's dropout sites;
    x is (B, L, C) on the padded rectangle of `space`."""

    def __init__(self, pseed: int, task: int = 0, probs: Optional[dict] = None):
        self.pseed, self.task = int(pseed), int(task)
        self.probs = dict(enc=0.2, dec=0.2, vp=0.5, postnet=0.5)    # config/model/base.yaml:10-11,16; Layers.py:133-134
        if probs:
            self.probs.update(probs)
        self.S = self.Tcap = None
        self.flen = self.foff = None
        self.cache = {}

    def bind(self, S: int, mel_lens: Optional[Sequence[int]], Tcap: Optional[int]):
        self.S = int(S)
        if mel_lens is not None:
            self.Tcap = int(Tcap)
            self.flen = [max(0, min(int(m), self.Tcap)) for m in mel_lens]     # engine.h: set_batches (frames kept per utterance)
            self.foff, o = [], G
            for n in self.flen:
                self.foff.append(o)
                o += n + G

    def rows(self, space: str, B: int, L: int) -> np.ndarray:
        """Engine row of every (b, l) of a padded (B, L) rectangle; -1 where the engine keeps no row (padded frames of F)."""
        b = np.arange(B)[:, None]
        l = np.arange(L)[None, :]
        if space == "P":
            assert L == self.S
            return (G + b * (self.S + G) + l).reshape(-1)
        if space == "R":
            assert L == self.Tcap
            return (G + b * (self.Tcap + G) + l).reshape(-1)
        assert space == "F"
        r = np.asarray(self.foff)[:, None] + l
        return np.where(l < np.asarray(self.flen)[:, None], r, -1).reshape(-1)

    def precompute(self, B: int, S: int, mel_lens: Sequence[int], T_max: int, *, enc_layers=4, dec_layers=6, d_model=256, vp_filter=256,
                   postnet_dim=512, postnet_layers=5, n_mel=80, max_seq_len=1000):
        """Generate every mask of a teacher-forced pass over a (B, S) / (B, T_max) batch up front (phoneme-level features), so that a
        TIMED oracle run (bench.py cpu_baseline) pays what the reference pays for dropout — one multiply per site — and not for this
        file's numpy hashing."""
        Tcap = min(int(T_max), max_seq_len)
        self.bind(S, list(mel_lens), Tcap)
        sites = [(2 * l + a, "P", S, d_model, self.probs["enc"]) for l in range(enc_layers) for a in (0, 1)]
        sites += [(64 + 2 * l + a, "F", Tcap, d_model, self.probs["dec"]) for l in range(dec_layers) for a in (0, 1)]
        sites += [(b + a, "P", S, vp_filter, self.probs["vp"]) for b in (128, 132, 136) for a in (0, 1)]
        sites += [(192 + i, "R", Tcap, postnet_dim if i < postnet_layers - 1 else n_mel, self.probs["postnet"]) for i in range(postnet_layers)]
        for site, space, L, C, prob in sites:
            if prob > 0.0:
                self.cache[(site, space, B, L, C)] = self.scale_mask(site, space, B, L, C, prob)
        return self

    def scale_mask(self, site: int, space: str, B: int, L: int, C: int, prob: float) -> torch.Tensor:
        hit = self.cache.get((site, space, B, L, C))
        if hit is not None:
            return hit
        rows = self.rows(space, B, L)
        live = rows >= 0
        keep = keep_mask(site_seed(self.pseed, site), self.task, np.where(live, rows, 0), C, prob) & live[:, None]
        scale = np.float32(1.0) / (np.float32(1.0) - np.float32(prob))              # engine.h: drop_spec (fp32 arithmetic)
        return torch.from_numpy((keep.astype(np.float32) * scale).reshape(B, L, C))

    def apply(self, x: torch.Tensor, site: int, space: str, prob: float) -> torch.Tensor:
        if prob <= 0.0:
            return x
        B, L, C = x.shape
        return x * self.scale_mask(site, space, B, L, C, prob).to(x.dtype)
