"""Episodic sampler / collate / feature-tree reader (meta_tts_amd/data.py) — CPU only."""
import json
import os

import numpy as np
import pytest

import __graft_entry__ as ge
from meta_tts_amd import data as D
from oracle_util import tiny_dims

from data_tree import PHONES, SPEAKERS, write_tree as _write_tree

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "collate.npz")


def _phone_ids(line_text):
    """Stand-in front-end for tests that do not use the golden: position in PHONES + 1."""
    return [PHONES.index(p) + 1 for p in line_text.strip("{}").split()]


def _dataset(root):
    return D.ConcatDataset([D.FeatureDataset(root, "train.txt", _phone_ids)])


def test_reader_and_reprocess_layout(tmp_path):
    _write_tree(str(tmp_path))
    ds = _dataset(str(tmp_path))
    assert len(ds) == sum(SPEAKERS.values())
    s = ds[13]
    assert s["id"] == "spkB_utt01" and s["speaker"] == 4 and s["mel"].shape[0] == s["duration"].sum()
    b = D.reprocess([ds[i] for i in range(4)], [2, 0, 3])
    assert len(b) == 12 and b[0] == ["spkA_utt02", "spkA_utt00", "spkA_utt03"]
    assert b[2].dtype == np.int64 and b[3].dtype == np.int64 and b[6].dtype == np.float32 and b[9].dtype == np.float32 and b[11].dtype == np.int64
    assert b[3].shape == (3, b[5]) and b[6].shape == (3, b[8], 32) and b[5] == b[4].max() and b[8] == b[7].max()
    for k in range(3):   # zero padding beyond each utterance
        assert np.all(b[3][k, b[4][k]:] == 0) and np.all(b[6][k, b[7][k]:] == 0) and np.all(b[11][k, b[4][k]:] == 0)
        assert b[11][k].sum() == b[7][k]


def test_val_tasks_are_fixed_and_persist(tmp_path):
    _write_tree(str(tmp_path))
    tasks = D.few_shot_task_dataset(_dataset(str(tmp_path)), ways=1, shots=2, queries=2, n_tasks_per_label=2, seed=1)
    assert len(tasks) == 4 and len(tasks.datasets) == 2           # spkC has only 3 < shots + queries samples
    first = [tasks[i] for i in range(4)]
    again = [tasks[i] for i in range(4)]
    for a, b in zip(first, again):
        assert a[0][0][0] == b[0][0][0] and a[1][0][0] == b[1][0][0]   # memoised episodes
    for sup, qry in first:
        ids = sup[0][0] + qry[0][0]
        assert len(sup) == 1 and len(qry) == 1 and len(sup[0][0]) == 2 and len(qry[0][0]) == 2
        assert len(set(ids)) == 4                                        # drawn without replacement
        assert len({i.split("_")[0] for i in ids}) == 1                  # one speaker per task
        assert len(set(sup[0][2].tolist() + qry[0][2].tolist())) == 1
    log = str(tmp_path / "log")
    m = D.prefetch_tasks(tasks, "val", log)
    assert sorted(m.values()) == [f"val_{i:03d}" for i in range(4)]
    fresh = D.few_shot_task_dataset(_dataset(str(tmp_path)), ways=1, shots=2, queries=2, n_tasks_per_label=2, seed=99)
    m2 = D.prefetch_tasks(fresh, "val", log)                             # recovers the persisted episodes
    assert m2 == m
    for i in range(4):
        assert fresh[i][0][0][0] == first[i][0][0][0] and fresh[i][1][0][0] == first[i][1][0][0]


def test_train_stream_feeds_the_engine(tmp_path):
    _write_tree(str(tmp_path))
    stream = D.few_shot_task_dataset(_dataset(str(tmp_path)), ways=1, shots=2, queries=2, seed=3)
    it = iter(stream)
    seen = set()
    for _ in range(12):
        sup, qry = next(it)
        spk = {i.split("_")[0] for i in sup[0][0] + qry[0][0]}
        assert len(spk) == 1
        seen |= spk
    assert "spkC" in seen or len(seen) >= 2      # every speaker is eligible in training (sampling with replacement)
    epoch = D.few_shot_task_dataset(_dataset(str(tmp_path)), ways=1, shots=2, queries=2, epoch_length=5, seed=3)
    assert len(epoch) == 5 and len([t for t in epoch]) == 5
    # the collated task is what the engine's boundary takes (reference 12-tuples)
    from meta_tts_amd.engine import Engine
    from meta_tts_amd import synth
    dims = tiny_dims()
    eng = Engine(dims, adapt_modules=["speaker_emb", "variance_adaptor", "decoder", "mel_linear", "postnet"], max_tasks=1, max_B=2,
                 max_S=16, max_T=96, lib_path=ge.build_emulator())
    eng.load_params(synth.make_params(dims, 0))
    sup, qry = next(it)
    eng.set_batches(0, [sup[0]])
    eng.set_batches(1, [qry[0]], spk_from=[sup[0]], average_spk=True)
    q, s = eng.meta_grad(1, 1e-3, 1.0)
    assert np.all(np.isfinite(q)) and np.all(np.isfinite(s))
    eng.close()



# ---------------------------------------------------------------------------------------------------------------------
# exact parity with the reference's reader + collate (tests/golden/collate.npz, made by importing dataset.py and
# lightning/collate.py in the build container): values AND dtypes of every element of the 12-tuples
# ---------------------------------------------------------------------------------------------------------------------
def _check(prefix, got, g):
    ids, raw, spk, texts, tlens, tmax, mels, mlens, mmax, pit, ene, dur = got
    assert list(ids) == [str(x) for x in g[prefix + "ids"]]
    assert list(raw) == [str(x) for x in g[prefix + "raw"]]
    for k, v in (("spk", spk), ("texts", texts), ("tlens", tlens), ("mels", mels), ("mlens", mlens), ("pit", pit), ("ene", ene), ("dur", dur)):
        a = np.asarray(v)
        np.testing.assert_array_equal(a, g[prefix + k], err_msg=prefix + k)
        assert str(a.dtype) == str(g[prefix + k + "_dtype"]), (prefix + k, a.dtype, g[prefix + k + "_dtype"])
    assert int(tmax) == int(g[prefix + "tmax"]) and int(mmax) == int(g[prefix + "mmax"])


def test_collate_matches_reference_golden_exactly(tmp_path):
    g = np.load(GOLDEN, allow_pickle=False)
    lines = _write_tree(str(tmp_path))
    # the reference's text front-end (text/__init__.py, out of scope) is replaced by its recorded output per line
    ids_of = {ln.split("|")[2]: g[f"text_{i}"].tolist() for i, ln in enumerate(lines)}
    ds = D.FeatureDataset(str(tmp_path), "train.txt", lambda t: ids_of[t])
    assert len(ds) == int(g["n"])
    samples = [ds[i] for i in range(len(ds))]
    for i, s in enumerate(samples):
        assert s["speaker"] == int(g[f"speaker_{i}"])
    _check("re_", D.reprocess(samples, [2, 0, 13, 22]), g)
    _check("single_", D.get_single_collate(sort=True)([samples[i] for i in (5, 1, 9, 20, 14)]), g)
    task = [samples[i] for i in (12, 15, 13, 19, 17, 14)]
    for sort in (False, True):
        sup, qry = D.SpeakerTaskCollate().meta_collate_fn(task, shots=3, queries=3, sort=sort, split=True)
        assert len(sup) == 1 and len(qry) == 1
        _check(f"meta{int(sort)}_sup_", sup[0], g)
        _check(f"meta{int(sort)}_qry_", qry[0], g)
    whole = D.SpeakerTaskCollate().meta_collate_fn(task, shots=3, queries=3, sort=False, split=False)
    _check("nosplit_", whole[0], g)
    _check("sub_", D.split_reprocess(whole[0], [4, 1]), g)


def test_ref_mel_speaker_args_match_reference_golden(tmp_path):
    """dvec / encoder speaker modes: `speaker_args` = (ref_mels, ref_slices) from the samples' spk_ref_mel_slices
    (dataset.py:83-91, collate.py:29-43) and its re-slicing for a sub-batch (collate.py:84-94)."""
    g = np.load(GOLDEN, allow_pickle=False)
    lines = _write_tree(str(tmp_path))
    ids_of = {ln.split("|")[2]: g[f"text_{i}"].tolist() for i, ln in enumerate(lines)}
    ds = D.FeatureDataset(str(tmp_path), "train.txt", lambda t: ids_of[t], spk_refer_wav=True)
    samples = [ds[i] for i in range(len(ds))]
    rb = D.reprocess(samples, [2, 0, 13, 22])
    for prefix, b in (("ref_", rb), ("refsub_", D.split_reprocess(rb, [3, 1]))):
        mels, slices = b[2]
        assert mels.dtype == np.float32
        np.testing.assert_array_equal(mels, g[prefix + "refmels"])
        assert [[s.start, s.stop] for s in slices] == g[prefix + "refbounds"].tolist()
        _check(prefix, b[:2] + (np.zeros(0),) + b[3:], g)

