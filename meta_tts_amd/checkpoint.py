"""Checkpoint layout compatible with the reference's PyTorch-Lightning files:
``{"state_dict": {"model.<name>": tensor, "learner.module.<name>": same tensor, ...}, "global_step": int,
"optimizer_states": [...]}`` written with plain ``torch.save``.  ``load_checkpoint`` applies the loader
surgery of lightning/systems/system.py:115-192 (old ``model.speaker_emb.weight`` key, 326 <-> 2390 speaker
tables, unknown keys dropped, optimizer state discarded when anything changed)."""
from __future__ import annotations

from typing import Dict

import numpy as np


def save_checkpoint(system, path: str):
    import torch
    sd = {k: (torch.tensor(v.item()) if np.ndim(v) == 0 else torch.from_numpy(np.ascontiguousarray(v)))
          for k, v in system.state_dict().items()}
    eng = system.engine
    opt = {"step": int(system.global_step),
           "exp_avg": {n: torch.from_numpy(eng.export(n, 4)) for n in eng.params},
           "exp_avg_sq": {n: torch.from_numpy(eng.export(n, 5)) for n in eng.params}}
    torch.save({"state_dict": sd, "global_step": int(system.global_step), "optimizer_states": [opt],
                "hyper_parameters": {"algorithm_config": system.algorithm_config, "model_config": system.model_config}}, path)


def adapt_state_dict(state_dict: Dict, model_sd: Dict, dataset: str = "LibriTTS", avg_train_spk_emb: bool = False):
    """system.py:121-192 restated on plain dicts.  Returns (new_state_dict, changes, is_changed)."""
    sd = dict(state_dict)
    changes = {"skip": [], "drop": [], "replace": [], "miss": []}
    changed = False
    if "model.speaker_emb.weight" in sd:  # old key (system.py:122-128)
        assert "model.speaker_emb.model.weight" not in sd
        sd["model.speaker_emb.model.weight"] = sd.pop("model.speaker_emb.weight")
        changes["replace"].append(["model.speaker_emb.weight", "model.speaker_emb.model.weight"])
        changed = True
    for k in list(sd):
        if k in model_sd:
            a, b = np.asarray(sd[k]), np.asarray(model_sd[k])
            if a.shape != b.shape:
                if k == "model.speaker_emb.model.weight":
                    new = b.copy()
                    if dataset == "LibriTTS":  # train-clean-100 era table -> all-subsets table (system.py:137-148)
                        assert a.shape[0] == 326 and b.shape[0] == 2390, (a.shape, b.shape)
                        new[:247] = a[:247]
                        new[-79:] = a[-79:]
                    elif avg_train_spk_emb:
                        assert a.shape[0] in (326, 2390)
                        new[:] = a[:247].mean(axis=0)
                    sd[k] = new
                else:
                    sd[k] = b
                changes["skip"].append([k, b.shape, a.shape])
                changed = True
        else:
            changes["drop"].append(k)
            del sd[k]
            changed = True
    for k in model_sd:
        if k not in sd:
            changes["miss"].append(k)
            changed = True
    return sd, changes, changed


def load_checkpoint(system, path: str, strict: bool = False):
    import torch
    ckpt = torch.load(path, map_location="cpu", weights_only=False)
    model_sd = system.state_dict()
    raw = {k: (v.numpy() if hasattr(v, "numpy") else np.asarray(v)) for k, v in ckpt["state_dict"].items()}
    sd, changes, changed = adapt_state_dict(raw, model_sd, system.preprocess_config.get("dataset", "LibriTTS"),
                                            system.algorithm_config["adapt"]["test"].get("avg_train_spk_emb", False))
    system.model.load_state_dict({k[len("model."):]: v for k, v in sd.items() if k.startswith("model.")}, strict=strict)
    system.global_step = int(ckpt.get("global_step", 0))
    system.test_global_step = system.global_step
    if changed:
        ckpt.pop("optimizer_states", None)  # system.py:191-192
    opt = (ckpt.get("optimizer_states") or [None])[0]
    if opt and "exp_avg" in opt:
        for n in system.engine.params:
            system.engine.import_state(n, 4, opt["exp_avg"][n].numpy())
            system.engine.import_state(n, 5, opt["exp_avg_sq"][n].numpy())
        system.engine.set_optimizer_step(int(opt.get("step", system.global_step)))
    return changes
