"""Checkpoint layout compatible with the reference's PyTorch-Lightning files:
``{"state_dict": {"model.<name>": tensor, "learner.module.<name>": same tensor, ...}, "global_step": int,
"optimizer_states": [torch Adam state_dict], "lr_schedulers": [LambdaLR state_dict], "epoch": int}``
written with plain ``torch.save`` — the optimizer / scheduler entries are exactly what ``optimizer.load_state_dict`` /
``scheduler.load_state_dict`` of the reference accept (params indexed in ``model.parameters()`` order, torch layouts).  ``load_checkpoint`` applies the loader
surgery of lightning/systems/system.py:115-192 (old ``model.speaker_emb.weight`` key, 326 <-> 2390 speaker
tables, unknown keys dropped, optimizer state discarded when anything changed)."""
from __future__ import annotations

from typing import Dict

import numpy as np


def _param_order(system):
    """``model.parameters()`` order of the reference FastSpeech2 (= its state_dict order minus the BatchNorm buffers), frozen
    nn.Parameters included: torch's Adam indexes its state by position in this list (lightning/optimizer.py:9-15 passes
    ``model.parameters()``); the four frozen tables (position_enc x2, pitch_bins, energy_bins) sit in the param group but never
    receive a gradient, so they have no state entry."""
    from . import synth
    order = [(n, trainable) for n, (_, trainable) in synth.param_spec(system.model.dims).items()]
    enc = getattr(system.model, "speaker_encoder", None)
    if enc is not None:   # speaker_emb is an LSTM encoder, not a table: its nn.LSTM / nn.Linear parameters take the table's place
        from .speaker_encoder import tensor_shapes
        trained = getattr(system.model, "spk_mode", "dvec") != "dvec"   # dvec: frozen (speaker_encoder.py:58), listed but stateless
        order = [(n, t) for n, t in order if n != "speaker_emb.model.weight"]
        order += [("speaker_emb.model." + n, trained) for n in tensor_shapes(**enc.cfg)]
    return order


def _enc_name(system, n):
    """The speaker encoder's own tensor name when parameter `n` belongs to it (speaker_emb.model.{lstm,linear}.*), else None."""
    pre = "speaker_emb.model."
    if getattr(system.model, "speaker_encoder", None) is not None and n.startswith(pre) and n != pre + "weight":
        return n[len(pre):]
    return None


def optimizer_state_dict(system) -> Dict:
    """``torch.optim.Adam.state_dict()`` of the reference's optimizer, filled from the engine's moments."""
    import torch
    from .systems import noam_lr
    eng, o = system.engine, system.train_config["optimizer"]
    init_lr = float(system.model.dims.d_model ** -0.5)
    steps = int(eng_adam_steps(system))
    state = {}
    order = _param_order(system)
    if steps > 0:
        for i, (n, trainable) in enumerate(order):
            if trainable and _enc_name(system, n):
                enc = system.model.speaker_encoder
                state[i] = {"step": torch.tensor(float(steps)), "exp_avg": torch.from_numpy(enc.export(_enc_name(system, n), 2)),
                            "exp_avg_sq": torch.from_numpy(enc.export(_enc_name(system, n), 3))}
            elif trainable:
                state[i] = {"step": torch.tensor(float(steps)), "exp_avg": torch.from_numpy(eng.export(n, 4)),
                            "exp_avg_sq": torch.from_numpy(eng.export(n, 5))}
    group = {"lr": noam_lr(system.global_step, system.model.dims.d_model, system.train_config), "betas": tuple(o["betas"]), "eps": o["eps"],
             "weight_decay": o["weight_decay"], "amsgrad": False, "maximize": False, "foreach": None, "capturable": False,
             "differentiable": False, "fused": None, "initial_lr": init_lr, "params": list(range(len(order)))}
    return {"state": state, "param_groups": [group]}


def eng_adam_steps(system) -> int:
    return int(getattr(system, "adam_steps", system.global_step))


def scheduler_state_dict(system) -> Dict:
    """``LambdaLR.state_dict()`` (lightning/scheduler.py:25-28): a function lambda is stored as None."""
    from .systems import noam_lr
    init_lr = float(system.model.dims.d_model ** -0.5)
    return {"base_lrs": [init_lr], "last_epoch": int(system.global_step), "_step_count": int(system.global_step) + 1,
            "_get_lr_called_within_step": False, "_last_lr": [noam_lr(system.global_step, system.model.dims.d_model, system.train_config)],
            "lr_lambdas": [None]}


def save_checkpoint(system, path: str):
    import torch
    sd = {k: (torch.tensor(v.item()) if np.ndim(v) == 0 else torch.from_numpy(np.ascontiguousarray(v)))
          for k, v in system.state_dict().items()}
    torch.save({"epoch": int(getattr(system, "current_epoch", 0)), "global_step": int(system.global_step),
                "pytorch-lightning_version": "1.x (written by meta_tts_amd)", "state_dict": sd,
                "optimizer_states": [optimizer_state_dict(system)], "lr_schedulers": [scheduler_state_dict(system)],
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
    system.adam_steps = 0
    if opt:
        if "state" in opt and "param_groups" in opt:   # torch.optim.Adam.state_dict(), what Lightning writes
            order = _param_order(system)
            if len(opt["param_groups"][0]["params"]) != len(order):
                raise ValueError(f"optimizer state covers {len(opt['param_groups'][0]['params'])} parameters, the model has {len(order)}")
            step = 0
            for i, (n, trainable) in enumerate(order):
                st = opt["state"].get(i)
                if st is None:
                    continue
                if not trainable:
                    raise ValueError(f"optimizer state for the frozen parameter {n}")
                if _enc_name(system, n):
                    system.model.speaker_encoder.import_state(_enc_name(system, n), 2, np.asarray(st["exp_avg"]))
                    system.model.speaker_encoder.import_state(_enc_name(system, n), 3, np.asarray(st["exp_avg_sq"]))
                else:
                    system.engine.import_state(n, 4, np.asarray(st["exp_avg"]))
                    system.engine.import_state(n, 5, np.asarray(st["exp_avg_sq"]))
                step = max(step, int(float(st["step"])))
            system.engine.set_optimizer_step(step)
            if getattr(system.model, "speaker_encoder", None) is not None and getattr(system.model, "spk_mode", "dvec") != "dvec":
                system.model.speaker_encoder.set_optimizer_step(step)
            system.adam_steps = step
        else:
            import warnings
            warnings.warn("unrecognised optimizer_states layout: Adam moments reset (the step count restarts the bias correction)")
            _reset_optimizer(system)
    else:
        _reset_optimizer(system)  # system.py:191-192 drops the optimizer state when the loader changed anything
    return changes


def _reset_optimizer(system):
    """Fresh Adam state for EVERY trained parameter: the engine's moments and, for speaker_emb: encoder / scratch_encoder, the LSTM
    speaker encoder's moments and step count (left as they were they would resume with stale moments under a restarted bias
    correction, and a later save would write moments inconsistent with the recorded step)."""
    system.engine.reset_optimizer()
    enc = getattr(system.model, "speaker_encoder", None)
    if enc is not None and getattr(system.model, "spk_mode", "dvec") in ("encoder", "scratch_encoder"):
        from .speaker_encoder import tensor_shapes
        for name, shape in tensor_shapes(**enc.cfg).items():
            zeros = np.zeros(shape, np.float32)
            enc.import_state(name, 2, zeros)
            enc.import_state(name, 3, zeros)
        enc.set_optimizer_step(0)
