#!/usr/bin/env python3
"""The reference's config surface as data (BUILD-CONTAINER ONLY: reads /root/reference/config and imports the reference model).

For every config/algorithm/*.yaml (30 files) and the train YAMLs main.py merges (config/train/base.yaml + {LibriTTS, VCTK, miniLibriTTS, dev}.yaml,
main.py:178-190) this writes the PARSED content (yaml.safe_load -> JSON: data, not source) and what the REFERENCE does with each algorithm file:

  reference_model_ctor   "ok", or the exception FastSpeech2(preprocess, model, algorithm) raises (lightning/model/fastspeech2.py:23-38,
                         speaker_encoder.py:46-60, phoneme_embedding.py:35-46) — e.g. the legacy-schema files have no adapt.type: KeyError('type')
  reference_system_keys  "ok", or the first key BaseAdaptorSystem.__init__ cannot read (base_adaptor.py:29-38: adapt.task.lr, adapt.modules,
                         adapt.train.steps, adapt.test.steps) — evaluated on the parsed dict (learn2learn is not installed, so the class itself
                         cannot be constructed here)

Usage:  python tests/golden/make_config_golden.py      (writes tests/golden/config_surface.json)
"""
import glob
import json
import os
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402
from meta_tts_amd.config import SYNTH_N_SPEAKER, SYNTH_STATS, load_yaml  # noqa: E402

REF = "/root/reference"


def model_ctor(alg):
    MG.install_shims()
    if REF not in sys.path:
        sys.path.insert(0, REF)
    pre = load_yaml(os.path.join(REF, "config/preprocess/LibriTTS.yaml"))
    mod = load_yaml(os.path.join(REF, "config/model/base.yaml"))
    tmp = tempfile.mkdtemp(prefix="mtts_cfg_")
    with open(os.path.join(tmp, "stats.json"), "w") as f:
        json.dump(SYNTH_STATS, f)
    with open(os.path.join(tmp, "speakers.json"), "w") as f:
        json.dump({str(i): i for i in range(SYNTH_N_SPEAKER)}, f)
    pre["path"]["preprocessed_path"] = tmp
    from lightning.model.fastspeech2 import FastSpeech2
    try:
        m = FastSpeech2(pre, mod, alg)
        return "ok", sorted({k.split(".")[0] for k, _ in m.named_parameters()})
    except Exception as ex:  # noqa: BLE001
        return f"{type(ex).__name__}: {ex}", None


def system_keys(alg):
    try:
        alg["adapt"]["task"]["lr"]
        alg["adapt"]["modules"]
        alg["adapt"]["train"]["steps"]
        alg["adapt"]["test"]["steps"]
        return "ok"
    except KeyError as ex:
        return f"KeyError: {ex}"


def main():
    out = {"algorithm": {}, "train": {}}
    for path in sorted(glob.glob(os.path.join(REF, "config/algorithm/*.yaml"))):
        alg = load_yaml(path)
        ctor, tops = model_ctor(load_yaml(path))
        out["algorithm"][os.path.basename(path)] = {"parsed": alg, "reference_model_ctor": ctor, "reference_model_top_level_modules": tops,
                                                    "reference_system_keys": system_keys(alg)}
    for path in sorted(glob.glob(os.path.join(REF, "config/train/*.yaml"))):
        out["train"][os.path.basename(path)] = load_yaml(path)
    with open(os.path.join(HERE, "config_surface.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    for k, v in out["algorithm"].items():
        print(k, "|", v["parsed"].get("type"), "|", v["reference_model_ctor"][:60], "|", v["reference_system_keys"])


if __name__ == "__main__":
    main()
