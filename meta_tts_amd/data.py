"""Episodic task sampling + collate for the hot path's input side — host code, numpy only.

Mirrors the reference's data interface so real LibriTTS features can replace the synthetic batches:

* `FeatureDataset`     — dataset.py:12-109: `<preprocessed>/<txt>` lines `basename|speaker|{phones}|raw text`,
                         `speakers.json`, `<kind>/<speaker>-<kind>-<basename>.npy` for mel / pitch / energy / duration;
                         items are the reference's sample dicts (id, speaker, text, raw_text, mel, pitch, energy, duration).
                         The phoneme -> id front-end (text/__init__.py) is out of scope: pass `text_to_sequence`.
* `reprocess`          — lightning/collate.py:9-60: pad + stack a list of samples into the 12-tuple every system consumes.
* `SpeakerTaskCollate` — lightning/collate.py:146-196: 1-way task of K+Q samples -> ([support 12-tuple], [query 12-tuple]),
                         the first `shots` samples are the support set.
* `few_shot_task_dataset` — lightning/datamodules/utils.py:14-65 (learn2learn MetaDataset / FusedNWaysKShots / TaskDataset,
                         un-vendored): train = endless stream of tasks, one random speaker per task, K+Q samples drawn WITH
                         replacement; val/test = `n_tasks_per_label` fixed tasks for every speaker with >= K+Q samples, drawn
                         WITHOUT replacement and memoised per task index (a task keeps its samples once drawn).
* `write_descriptions / load_descriptions / prefetch_tasks` — datamodules/utils.py:67-126: persist the sampled val/test tasks
                         (`{tag}_descriptions.json`, `{tag}_SQids.json`) so every run evaluates the same episodes.

Arrays are numpy (the reference returns torch CPU tensors of the same dtypes); `meta_tts_amd.engine.Engine.set_batches` takes
either.
"""
from __future__ import annotations

import json
import os
from typing import Callable, Dict, List, Optional, Sequence

import numpy as np


def pad_1D(xs: Sequence[np.ndarray], pad: float = 0) -> np.ndarray:          # utils/tools.py:262-274
    n = max(len(x) for x in xs)
    return np.stack([np.pad(x, (0, n - len(x)), mode="constant", constant_values=pad) for x in xs])


def pad_2D(xs: Sequence[np.ndarray]) -> np.ndarray:                          # utils/tools.py:277-294
    n = max(x.shape[0] for x in xs)
    return np.stack([np.pad(x, ((0, n - x.shape[0]), (0, 0)), mode="constant") for x in xs])


def _ref_mel_args(ref_mels):
    """collate.py:29-43: concatenate the utterances' partial-utterance stacks, one consecutive slice per utterance."""
    slices, start = [], 0
    for m in ref_mels:
        slices.append(slice(start, start + m.shape[0]))
        start += m.shape[0]
    return np.concatenate(ref_mels, axis=0).astype(np.float32), slices


def reprocess(data: Sequence[dict], idxs: Sequence[int]):
    """lightning/collate.py:9-60.  `speaker_args` = int64 speaker ids, or — when the samples carry `spk_ref_mel_slices` (the dvec /
    encoder speaker modes, dataset.py:83-91) — the pair (ref_mels (n_partials, 160, 40), [slice per utterance])."""
    pick = [data[i] for i in idxs]
    texts = [np.asarray(d["text"]) for d in pick]
    mels = [np.asarray(d["mel"], np.float32) for d in pick]
    text_lens = np.array([t.shape[0] for t in texts])
    mel_lens = np.array([m.shape[0] for m in mels])
    if "spk_ref_mel_slices" in data[0]:
        speaker_args = _ref_mel_args([np.asarray(d["spk_ref_mel_slices"]) for d in pick])
    else:
        speaker_args = np.array([d["speaker"] for d in pick], np.int64)
    return ([d["id"] for d in pick], [d["raw_text"] for d in pick],
            speaker_args,
            pad_1D(texts).astype(np.int64), text_lens, int(text_lens.max()),
            pad_2D(mels).astype(np.float32), mel_lens, int(mel_lens.max()),
            pad_1D([np.asarray(d["pitch"]) for d in pick]).astype(np.float32),
            pad_1D([np.asarray(d["energy"]) for d in pick]),
            pad_1D([np.asarray(d["duration"]) for d in pick]).astype(np.int64))


def split_reprocess(batch, idxs):
    """lightning/collate.py:63-127 (table-speaker path): the sub-batch `idxs` of a collated 12-tuple, re-cropped to its own
    maximum lengths (phoneme- or frame-level pitch / energy detected by their padded width, as the reference does)."""
    ids, raw, spk, texts, tlens, tmax, mels, mlens, mmax, pit, ene, dur = batch
    idxs = np.asarray(idxs)
    A = lambda x: np.asarray(x)
    tl, ml = A(tlens)[idxs], A(mlens)[idxs]
    st, sm = int(tl.max()), int(ml.max())
    crop = lambda x: A(x)[idxs][:, :st] if A(x).shape[1] == int(tmax) else A(x)[idxs][:, :sm]
    if isinstance(spk, tuple):   # (ref_mels, ref_slices): collate.py:84-94
        sub_spk = _ref_mel_args([A(spk[0])[spk[1][i]] for i in idxs])
    else:
        sub_spk = A(spk)[idxs]
    return ([ids[i] for i in idxs], [raw[i] for i in idxs], sub_spk, A(texts)[idxs][:, :st], tl, st,
            A(mels)[idxs][:, :sm], ml, sm, crop(pit), crop(ene), A(dur)[idxs][:, :st])


class Task:
    """lightning/systems/utils.py:80-117: iterate a task's support set in mini-batches (drop_last), e.g. batch_size = 1 for
    the 1-shot test mode (base_adaptor.py:139-147).  shuffle uses its own RandomState."""

    def __init__(self, sup_data, qry_data, batch_size=None, shuffle=True, seed=0):
        self.sup_data, self.qry_data, self.batch_size, self.shuffle = sup_data, qry_data, batch_size, shuffle
        self.rng = np.random.RandomState(seed)
        self.reset_iterator()

    def reset_iterator(self):
        n = len(self.sup_data[0])
        order = self.rng.permutation(n) if self.shuffle else np.arange(n)
        bs = self.batch_size or n
        self._batches = [order[i:i + bs] for i in range(0, n - bs + 1, bs)]
        self._pos = 0

    def next_batch(self):
        if self._pos >= len(self._batches):
            self.reset_iterator()
        idxs = self._batches[self._pos]
        self._pos += 1
        return split_reprocess(self.sup_data, idxs)

    def __iter__(self):
        self.reset_iterator()
        return self

    def __next__(self):
        if self._pos >= len(self._batches):
            raise StopIteration
        idxs = self._batches[self._pos]
        self._pos += 1
        return split_reprocess(self.sup_data, idxs)


def get_single_collate(sort: bool = True):
    """lightning/collate.py:130-143 (BaselineDataModule): one plain batch, longest text first when `sort`."""
    def collate_fn(data):
        idx = np.argsort(-np.array([d["text"].shape[0] for d in data])) if sort else np.arange(len(data))
        return reprocess(data, idx)
    return collate_fn


class SpeakerTaskCollate:
    """lightning/collate.py:146-196."""

    def get_meta_collate(self, shots: int, queries: int, sort: bool = False, split: bool = True):
        return lambda data: self.meta_collate_fn(data, shots, queries, sort, split)

    def meta_collate_fn(self, data, shots, queries, sort=False, split=True):
        batch_size = shots + queries
        assert len(data) == batch_size, "n_batch=1 for speaker adaptation"
        idx = np.argsort(-np.array([d["text"].shape[0] for d in data])) if sort else np.arange(batch_size)
        idx = idx.reshape((-1, batch_size))
        if not split:
            return [reprocess(data, row) for row in idx]
        sup = np.zeros(batch_size, dtype=bool)
        sup[:shots] = True
        return ([reprocess(data, row) for row in idx[:, sup]], [reprocess(data, row) for row in idx[:, ~sup]])


class FeatureDataset:
    """dataset.py:12-109 reader of the preprocessed feature tree (see module docstring)."""

    def __init__(self, preprocessed_path: str, filename: str, text_to_sequence: Callable[[str], Sequence[int]], spk_refer_wav: bool = False):
        self.root = preprocessed_path
        self.text_to_sequence = text_to_sequence
        self.spk_refer_wav = spk_refer_wav   # dataset.py:16,83-91: also read spk_ref_mel_slices/{spk}-mel-{basename}.npy
        self.basename, self.speaker, self.text, self.raw_text = [], [], [], []
        with open(os.path.join(preprocessed_path, filename), encoding="utf-8") as f:
            for line in f:
                n, s, t, r = line.strip("\n").split("|")
                self.basename.append(n); self.speaker.append(s); self.text.append(t); self.raw_text.append(r)
        with open(os.path.join(preprocessed_path, "speakers.json")) as f:
            self.speaker_map = json.load(f)

    def __len__(self):
        return len(self.text)

    def _load(self, kind, idx):
        return np.load(os.path.join(self.root, kind, f"{self.speaker[idx]}-{kind}-{self.basename[idx]}.npy"))

    def __getitem__(self, idx):
        sample = {"id": self.basename[idx], "speaker": self.speaker_map[self.speaker[idx]],
                  "text": np.array(self.text_to_sequence(self.text[idx])), "raw_text": self.raw_text[idx],
                  "mel": self._load("mel", idx), "pitch": self._load("pitch", idx), "energy": self._load("energy", idx),
                  "duration": self._load("duration", idx)}
        if self.spk_refer_wav:
            sample["spk_ref_mel_slices"] = np.load(os.path.join(self.root, "spk_ref_mel_slices", f"{self.speaker[idx]}-mel-{self.basename[idx]}.npy"))
        return sample


class ConcatDataset:
    """torch.utils.data.ConcatDataset semantics for the few datasets a run concatenates (datamodules/*.py)."""

    def __init__(self, datasets):
        self.datasets = list(datasets)
        self.cum = np.cumsum([len(d) for d in self.datasets])

    def __len__(self):
        return int(self.cum[-1]) if len(self.cum) else 0

    def __getitem__(self, idx):
        k = int(np.searchsorted(self.cum, idx, side="right"))
        return self.datasets[k][idx - (int(self.cum[k - 1]) if k else 0)]


def get_multispeaker_id2lb(datasets) -> Dict[int, str]:
    """datamodules/utils.py:129-137: global sample index -> speaker label over the concatenated datasets."""
    id2lb, total = {}, 0
    for ds in datasets:
        for i, spk in enumerate(ds.speaker):
            id2lb[total + i] = spk
        total += len(ds)
    return id2lb


class TaskDataset:
    """One family of 1-way tasks over `dataset` (learn2learn TaskDataset + FusedNWaysKShots + LoadData restated).

    labels: the speaker labels tasks may use; num_tasks = -1: endless stream, each access draws a fresh task with
    replacement; num_tasks > 0: that many tasks, drawn without replacement on first access and then fixed
    (`sampled_descriptions[i]` = the sample indices of task i, what write_descriptions persists)."""

    def __init__(self, dataset, labels_to_indices: Dict[str, List[int]], labels: Sequence[str], k: int, num_tasks: int,
                 task_collate, replacement: bool, rng: np.random.RandomState):
        self.dataset, self.l2i, self.labels, self.k = dataset, labels_to_indices, list(labels), k
        self.num_tasks, self.task_collate, self.replacement, self.rng = num_tasks, task_collate, replacement, rng
        self.sampled_descriptions: Dict[int, List[int]] = {}

    def __len__(self):
        return self.num_tasks if self.num_tasks > 0 else 1

    def sample_indices(self) -> List[int]:
        label = self.labels[int(self.rng.randint(len(self.labels)))]
        pool = self.l2i[label]
        return [int(i) for i in self.rng.choice(pool, size=self.k, replace=self.replacement)]

    def __getitem__(self, i):
        if self.num_tasks <= 0:
            idxs = self.sample_indices()
        else:
            if not 0 <= i < self.num_tasks:
                raise IndexError(i)
            if i not in self.sampled_descriptions:
                self.sampled_descriptions[i] = self.sample_indices()
            idxs = self.sampled_descriptions[i]
        return self.task_collate([self.dataset[j] for j in idxs])

    def __iter__(self):
        i = 0
        while self.num_tasks <= 0 or i < self.num_tasks:
            yield self[i]
            i += 1


def few_shot_task_dataset(dataset: ConcatDataset, ways: int, shots: int, queries: int, n_tasks_per_label: int = -1,
                          epoch_length: int = -1, seed: int = 0):
    """datamodules/utils.py:14-65 (speaker tasks).  Returns a TaskDataset (train) or a ConcatDataset of per-speaker
    TaskDatasets (val/test).  `epoch_length` > 0 bounds the train stream per epoch like EpisodicBatcher."""
    assert ways == 1, "the reference only builds 1-way (single-speaker) tasks"
    id2lb = get_multispeaker_id2lb(dataset.datasets)
    l2i: Dict[str, List[int]] = {}
    for i in range(len(dataset)):
        l2i.setdefault(id2lb[i], []).append(i)
    collate = SpeakerTaskCollate().get_meta_collate(shots, queries)
    rng = np.random.RandomState(seed)
    if n_tasks_per_label > 0:
        tasks = [TaskDataset(dataset, l2i, [lb], shots + queries, n_tasks_per_label, collate, False, rng)
                 for lb, idx in l2i.items() if len(idx) >= shots + queries]
        return ConcatDataset(tasks)
    t = TaskDataset(dataset, l2i, sorted(l2i), shots + queries, epoch_length if epoch_length > 0 else -1, collate, True, rng)
    if epoch_length > 0:   # an epoch = epoch_length fresh tasks (EpisodicBatcher): do not memoise
        t.__class__ = _EpochTaskDataset
    return t


class _EpochTaskDataset(TaskDataset):
    def __getitem__(self, i):
        if not 0 <= i < self.num_tasks:
            raise IndexError(i)
        return self.task_collate([self.dataset[j] for j in self.sample_indices()])


def write_descriptions(tasks: ConcatDataset, filename: str):
    with open(filename, "w") as f:
        json.dump([{str(i): ds.sampled_descriptions[i] for i in sorted(ds.sampled_descriptions)} for ds in tasks.datasets], f, indent=4)


def load_descriptions(tasks: ConcatDataset, filename: str):
    with open(filename) as f:
        loaded = json.load(f)
    assert len(tasks.datasets) == len(loaded), "TaskDataset count mismatch"
    for ds, desc in zip(tasks.datasets, loaded):
        assert len(desc) == ds.num_tasks, "num_tasks mismatch"
        for j, idxs in desc.items():
            ds.sampled_descriptions[int(j)] = [int(x) for x in idxs]


def get_SQids2Tid(tasks, tag: str):
    sq, m = [], {}
    for i, task in enumerate(tasks):
        sup_ids, qry_ids = task[0][0][0], task[1][0][0]
        sq.append({"sup_id": sup_ids, "qry_id": qry_ids})
        m[f"{'-'.join(sup_ids)}.{'-'.join(qry_ids)}"] = f"{tag}_{i:03d}"
    return sq, m


def prefetch_tasks(tasks: ConcatDataset, tag: str = "val", log_dir: str = "") -> Dict[str, str]:
    """datamodules/utils.py:107-126: recover the persisted episodes, or draw them once and persist."""
    dpath, spath = os.path.join(log_dir, f"{tag}_descriptions.json"), os.path.join(log_dir, f"{tag}_SQids.json")
    if os.path.exists(dpath) and os.path.exists(spath):
        load_descriptions(tasks, dpath)
        with open(spath) as f:
            sq = json.load(f)
        return {f"{'-'.join(d['sup_id'])}.{'-'.join(d['qry_id'])}": f"{tag}_{i:03d}" for i, d in enumerate(sq)}
    os.makedirs(log_dir, exist_ok=True)
    sq, m = get_SQids2Tid(tasks, tag)
    with open(spath, "w") as f:
        json.dump(sq, f, indent=4)
    write_descriptions(tasks, dpath)
    return m
