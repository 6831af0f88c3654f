#!/usr/bin/env python3
"""Golden vectors for the input side of the hot path, produced by the reference's own reader and
collate: dataset.py:14-109 (TTSDataset.__getitem__), lightning/collate.py:9-60 (reprocess), :63-127 (split_reprocess),
:130-143 (get_single_collate), :146-196 (SpeakerTaskCollate.meta_collate_fn) on the synthetic feature tree of
tests/data_tree.py.  BUILD-CONTAINER ONLY (imports /root/reference); only data is stored: the phoneme ids the reference's
text front-end gives every line (so the test can feed meta_tts_amd.data the same ids) and the collated 12-tuples.

Usage: python tests/golden/make_collate_golden.py     (writes tests/golden/collate.npz)"""
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
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, HERE)

import data_tree  # noqa: E402
from make_golden import install_shims  # noqa: E402


def tup(prefix, b, out):
    """flatten one 12-tuple into npz entries"""
    ids, raw, spk, texts, tlens, tmax, mels, mlens, mmax, pit, ene, dur = b
    out[prefix + "ids"] = np.array(list(ids))
    out[prefix + "raw"] = np.array(list(raw))
    for k, v in (("spk", spk), ("texts", texts), ("tlens", tlens), ("mels", mels), ("mlens", mlens), ("pit", pit), ("ene", ene), ("dur", dur)):
        a = v.numpy() if hasattr(v, "numpy") else np.asarray(v)
        out[prefix + k] = a
        out[prefix + k + "_dtype"] = np.array(str(a.dtype))
    out[prefix + "tmax"] = np.array(int(tmax))
    out[prefix + "mmax"] = np.array(int(mmax))


def main():
    install_shims()
    m = types.ModuleType("resemblyzer.audio"); m.preprocess_wav = None; m.wav_to_mel_spectrogram = None
    sys.modules["resemblyzer.audio"] = m
    sys.modules["resemblyzer"].audio = m
    sys.path.insert(0, REF)
    from dataset import TTSDataset
    from lightning.collate import SpeakerTaskCollate, get_single_collate, reprocess, split_reprocess
    root = tempfile.mkdtemp(prefix="mtts_collate_")
    lines = data_tree.write_tree(root)
    pre = {"dataset": "LibriTTS", "path": {"preprocessed_path": root}, "preprocessing": {"text": {"text_cleaners": ["english_cleaners"]}}}
    trn = {"optimizer": {"batch_size": 4}}
    ds = TTSDataset("train.txt", pre, trn)
    out = {"n": np.array(len(ds))}
    samples = [ds[i] for i in range(len(ds))]
    for i, s in enumerate(samples):
        out[f"text_{i}"] = np.asarray(s["text"], np.int64)
        out[f"speaker_{i}"] = np.array(s["speaker"])
    tup("re_", reprocess(samples, [2, 0, 13, 22]), out)                       # mixed speakers, unsorted
    tup("single_", get_single_collate(sort=True)([samples[i] for i in (5, 1, 9, 20, 14)]), out)
    task = [samples[i] for i in (12, 15, 13, 19, 17, 14)]                     # one speaker (spkB), K = 3 + Q = 3
    for sort in (False, True):
        sup, qry = SpeakerTaskCollate().meta_collate_fn(task, shots=3, queries=3, sort=sort, split=True)
        assert len(sup) == 1 and len(qry) == 1
        tup(f"meta{int(sort)}_sup_", sup[0], out)
        tup(f"meta{int(sort)}_qry_", qry[0], out)
    whole = SpeakerTaskCollate().meta_collate_fn(task, shots=3, queries=3, sort=False, split=False)
    tup("nosplit_", whole[0], out)
    tup("sub_", split_reprocess(whole[0], [4, 1]), out)                       # 1-shot test mode re-crop (systems/utils.py:80-117)
    # dvec / encoder speaker modes: samples carry spk_ref_mel_slices and speaker_args becomes (ref_mels, ref_slices) (collate.py:29-43,84-94)
    ds_ref = TTSDataset("train.txt", pre, trn, spk_refer_wav=True)
    ref_samples = [ds_ref[i] for i in range(len(ds_ref))]

    def tup_ref(prefix, b):
        mels_, slices_ = b[2]
        out[prefix + "refmels"] = mels_.numpy()
        out[prefix + "refbounds"] = np.array([[s.start, s.stop] for s in slices_], np.int64)
        tup(prefix, b[:2] + (np.zeros(0),) + b[3:], out)
    rb = reprocess(ref_samples, [2, 0, 13, 22])
    tup_ref("ref_", rb)
    tup_ref("refsub_", split_reprocess(rb, [3, 1]))
    np.savez_compressed(os.path.join(HERE, "collate.npz"), **out)
    print("collate golden written:", len(out), "entries")


if __name__ == "__main__":
    main()
