"""The reference's config surface: every config/algorithm/*.yaml (30 files) x the train YAML families main.py merges
(main.py:178-190), as parsed data in tests/golden/config_surface.json together with what the REFERENCE does with each algorithm file
(tests/golden/make_config_golden.py imported lightning/model/fastspeech2.py and recorded `FastSpeech2(...)`'s outcome; the keys
BaseAdaptorSystem.__init__ reads, base_adaptor.py:29-38, evaluated on the parsed dict).  Here: `get_system(type)` constructs a system from each on
the SIMT emulator (tiny model dims), agrees with the reference on which files construct and on HOW the others fail (same exception, same key),
and the constructed systems carry the file's values (registry class, speaker mode, adapted modules, inner lr / steps, optimizer block)."""
import copy
import json
import os

import numpy as np
import pytest

import __graft_entry__ as ge
from oracle_util import synth, tiny_dims
from meta_tts_amd.engine import MttsError
from meta_tts_amd.systems import BaselineSystem, IMAMLSystem, MetaSystem, get_system

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "config_surface.json")) as _f:
    SURFACE = json.load(_f)
ALGS = sorted(SURFACE["algorithm"])
TRAINS = ["base.yaml", "LibriTTS.yaml"]
REGISTRY = {"meta": MetaSystem, "baseline": BaselineSystem, "imaml": IMAMLSystem}    # lightning/systems/__init__.py:5-9


@pytest.fixture(scope="module")
def emu_lib():
    return ge.build_emulator()


def _train_config(name):
    trn = copy.deepcopy(SURFACE["train"]["base.yaml"])
    if name != "base.yaml":
        trn.update(copy.deepcopy(SURFACE["train"][name]))       # main.py:186-188: the corpus file updates the base file key by key
    return trn


def _build(emu_lib, alg, trn):
    dims = tiny_dims()
    pre = copy.deepcopy(dims.preprocess_config)
    pre["path"] = {"preprocessed_path": "/nonexistent"}
    return get_system(alg["type"])(pre, dims.model_config, trn, alg, max_tasks=1, max_batch=3, max_src_len=16, max_mel_len=96, lib_path=emu_lib), dims


def test_fixture_covers_the_reference_directory():
    assert len(ALGS) == 30 and {"base.yaml", "LibriTTS.yaml", "VCTK.yaml", "miniLibriTTS.yaml", "dev.yaml"} <= set(SURFACE["train"])
    assert {a["parsed"]["type"] for a in SURFACE["algorithm"].values()} == {"meta", "baseline", "imaml"}


@pytest.mark.parametrize("train", TRAINS)
@pytest.mark.parametrize("name", ALGS)
def test_system_from_every_algorithm_file(emu_lib, name, train):
    rec = SURFACE["algorithm"][name]
    alg, trn = copy.deepcopy(rec["parsed"]), _train_config(train)
    if rec["reference_model_ctor"] != "ok":
        # legacy-schema files (no adapt.type / adapt.task: base_share_emb_va_d, meta_lingual, meta_share_emb_va_d): the reference's model
        # constructor raises KeyError('type') at fastspeech2.py:35 — the same exception with the same key here, nothing half-built
        kind, _, msg = rec["reference_model_ctor"].partition(": ")
        assert kind == "KeyError"
        with pytest.raises(KeyError) as ei:
            _build(emu_lib, alg, trn)
        assert repr(ei.value.args[0]) == msg
        return
    if alg["adapt"]["type"] == "lang":
        # dev.yaml: adapt.type lang + codebook phoneme embedding — constructs in the reference, out of scope here (the
        # codebook front-end is not on the hot path): a NAMED refusal, not a silent fallback to the plain embedding
        with pytest.raises(MttsError, match="lang"):
            _build(emu_lib, alg, trn)
        return
    assert rec["reference_system_keys"] == "ok"
    sysm, dims = _build(emu_lib, alg, trn)
    try:
        assert type(sysm) is REGISTRY[alg["type"]]
        assert sysm.model.spk_mode == alg["adapt"]["speaker_emb"]
        assert tuple(sysm.model.adapt_modules) == tuple(alg["adapt"]["modules"])
        assert sysm.adaptation_lr == alg["adapt"]["task"]["lr"]
        assert sysm.adaptation_steps == alg["adapt"]["train"]["steps"] and sysm.test_adaptation_steps == alg["adapt"]["test"]["steps"]
        assert sysm.train_config["optimizer"] == trn["optimizer"] and sysm.train_config["step"] == trn["step"]
        # the reference model's top-level modules (recorded by the generator) are the engine's parameter families
        mine = {k.split(".")[0] for k in sysm.engine.params}
        ref = set(rec["reference_model_top_level_modules"])
        if alg["adapt"]["speaker_emb"] in ("dvec", "encoder", "scratch_encoder"):
            ref.discard("speaker_emb")        # the LSTM speaker encoder lives beside the acoustic model here (speaker_encoder.DVectorEncoder)
        assert ref <= mine | {"speaker_emb"}, (ref, mine)
    finally:
        sysm.engine.close()


@pytest.mark.parametrize("name", ["meta_emb_vad.yaml", "base_emb_vad.yaml", "meta_table_emb_va_d.yaml", "base_emb_vad.train_clean.1-shot.yaml"])
def test_one_training_step_per_family(emu_lib, name):
    """A constructed system also STEPS: one training_step with the file's own inner lr / steps on a tiny task, finite losses, a non-zero outer
    gradient on an adapted tensor."""
    alg, trn = copy.deepcopy(SURFACE["algorithm"][name]["parsed"]), _train_config("LibriTTS.yaml")
    sysm, _ = _build(emu_lib, alg, trn)
    dims = sysm.engine.dims           # (vocabulary / speaker count as the system read them from its configs)
    kw = dict(n_mel=dims.n_mel, vocab=dims.vocab, s_range=(5, 13), d_range=(1, 6), first_len=12)
    sup, qry = synth.make_batch(61, 3, speaker=2, **kw), synth.make_batch(62, 2, speaker=2, **kw)
    sysm.engine.load_params(synth.make_params(dims, 0))
    if alg["type"] == "meta":
        q, s = sysm.meta_learn_tasks([(sup, qry)])
        assert s.shape[0] == alg["adapt"]["train"]["steps"]
    else:
        q = sysm.engine_plain_grad([sup])
    assert np.isfinite(np.asarray(q)).all()
    assert float(np.abs(sysm.engine.export("mel_linear.weight", 1)).max()) > 0
    sysm.engine.close()
