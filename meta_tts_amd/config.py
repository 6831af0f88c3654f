"""Config surface of the hot path.

The YAML key layout is the reference's (config/model/base.yaml:1-30,
config/train/base.yaml:1-16, config/algorithm/meta_emb_vad.yaml:1-38,
config/preprocess/LibriTTS.yaml), so a reference config directory can be
loaded unchanged with :func:`load_configs`.  The defaults below restate the
values of those four files for the north-star configuration (meta_emb_vad on
LibriTTS); they are what bench.py / tests use when no YAML is given.
"""
from __future__ import annotations

import copy
from typing import Any, Dict, Sequence

VOCAB_SIZE = 361  # len(text.symbols.symbols) + 1 (transformer/Models.py:40)


def default_model_config() -> Dict[str, Any]:
    return {
        "transformer": {
            "encoder_layer": 4,
            "encoder_head": 2,
            "encoder_hidden": 256,
            "decoder_layer": 6,
            "decoder_head": 2,
            "decoder_hidden": 256,
            "conv_filter_size": 1024,
            "conv_kernel_size": [9, 1],
            "encoder_dropout": 0.2,
            "decoder_dropout": 0.2,
        },
        "variance_predictor": {"filter_size": 256, "kernel_size": 3, "dropout": 0.5},
        "variance_embedding": {
            "pitch_quantization": "linear",
            "energy_quantization": "linear",
            "n_bins": 256,
        },
        "multi_speaker": True,
        "multi_lingual": True,
        "max_seq_len": 1000,
        "vocoder": {"model": "MelGAN", "speaker": "universal"},
    }


def default_train_config() -> Dict[str, Any]:
    return {
        "optimizer": {
            "batch_size": 80,
            "betas": [0.9, 0.98],
            "eps": 1e-9,
            "weight_decay": 0.0,
            "grad_clip_thresh": 1.0,
            "grad_acc_step": 1,
            "warm_up_step": 4000,
            "anneal_steps": [300000, 400000, 500000],
            "anneal_rate": 0.3,
        },
        "step": {
            "total_step": 100000,
            "log_step": 100,
            "synth_step": 1000,
            "val_step": 1000,
            "save_step": 1000,
        },
    }


def default_algorithm_config() -> Dict[str, Any]:
    task = {"ways": 1, "shots": 5, "queries": 5, "lr": 0.001}
    return {
        "name": "meta_emb_vad",
        "type": "meta",
        "adapt": {
            "type": "spk",
            "speaker_emb": "table",
            "phoneme_emb": {"type": "embedding", "refresh": False},
            "modules": ["speaker_emb", "variance_adaptor", "decoder", "mel_linear", "postnet"],
            "task": dict(task),
            "train": dict(task, steps=5, meta_batch_size=8),
            "test": dict(
                task,
                queries=1,
                steps=100,
                saving_steps=[5, 10, 20, 50, 100, 200, 400, 600, 800, 1000],
                avg_train_spk_emb=False,
            ),
        },
    }


def default_preprocess_config() -> Dict[str, Any]:
    return {
        "dataset": "LibriTTS",
        "path": {"preprocessed_path": "./preprocessed_data/LibriTTS"},
        "preprocessing": {
            "audio": {"sampling_rate": 22050, "max_wav_value": 32768.0},
            "stft": {"filter_length": 1024, "hop_length": 256, "win_length": 1024},
            "mel": {"n_mel_channels": 80, "mel_fmin": 0, "mel_fmax": None},
            "pitch": {"feature": "phoneme_level", "normalization": True},
            "energy": {"feature": "phoneme_level", "normalization": True},
        },
    }


# Synthetic stand-ins for preprocessed_data/<corpus>/{stats,speakers}.json
# (read by the reference at lightning/model/modules.py:41-46 and
# speaker_encoder.py:49-50); values fixed by SURVEY.md section 8(d).
SYNTH_STATS = {"pitch": [-2.0, 8.0, 0.0, 1.0], "energy": [-1.5, 7.0, 0.0, 1.0]}
SYNTH_N_SPEAKER = 2390


def load_yaml(path: str) -> Dict[str, Any]:
    import yaml

    with open(path, "r") as f:
        return yaml.safe_load(f)


def load_configs(preprocess: Sequence[str] | str, model: str, train: Sequence[str], algorithm: str):
    """Mirror of the reference CLI merge rule (main.py:178-190): the second
    train YAML updates the first; the other families are loaded as-is."""
    pre = load_yaml(preprocess if isinstance(preprocess, str) else preprocess[0])
    mod = load_yaml(model)
    trn = load_yaml(train[0])
    for extra in train[1:]:
        trn.update(load_yaml(extra))
    alg = load_yaml(algorithm)
    return pre, mod, trn, alg


class ModelDims:
    """Flat view of the sizes the device library needs (mirrors mtts_model_cfg in include/mtts.h)."""

    def __init__(self, model_config=None, preprocess_config=None, n_speaker=SYNTH_N_SPEAKER,
                 stats=None, vocab=VOCAB_SIZE):
        mc = copy.deepcopy(model_config or default_model_config())
        pc = copy.deepcopy(preprocess_config or default_preprocess_config())
        st = stats or SYNTH_STATS
        t = mc["transformer"]
        assert t["encoder_hidden"] == t["decoder_hidden"], "encoder/decoder hidden must match (speaker add)"
        levels = ("phoneme_level", "frame_level")
        assert pc["preprocessing"]["pitch"]["feature"] in levels and pc["preprocessing"]["energy"]["feature"] in levels
        self.pitch_frame_level = pc["preprocessing"]["pitch"]["feature"] == "frame_level"
        self.energy_frame_level = pc["preprocessing"]["energy"]["feature"] == "frame_level"
        assert mc["variance_embedding"]["pitch_quantization"] == "linear"
        assert mc["variance_embedding"]["energy_quantization"] == "linear"
        self.d_model = int(t["encoder_hidden"])
        self.enc_layers = int(t["encoder_layer"])
        self.dec_layers = int(t["decoder_layer"])
        self.enc_heads = int(t["encoder_head"])
        self.dec_heads = int(t["decoder_head"])
        self.d_ff = int(t["conv_filter_size"])
        self.k1, self.k2 = (int(k) for k in t["conv_kernel_size"])
        self.enc_dropout = float(t["encoder_dropout"])
        self.dec_dropout = float(t["decoder_dropout"])
        self.vp_filter = int(mc["variance_predictor"]["filter_size"])
        self.vp_kernel = int(mc["variance_predictor"]["kernel_size"])
        self.vp_dropout = float(mc["variance_predictor"]["dropout"])
        self.n_bins = int(mc["variance_embedding"]["n_bins"])
        self.max_seq_len = int(mc["max_seq_len"])
        self.n_mel = int(pc["preprocessing"]["mel"]["n_mel_channels"])
        self.vocab = int(vocab)
        self.n_speaker = int(n_speaker)
        self.pitch_min, self.pitch_max = float(st["pitch"][0]), float(st["pitch"][1])
        self.energy_min, self.energy_max = float(st["energy"][0]), float(st["energy"][1])
        # PostNet sizes are hard-coded in the reference (transformer/Layers.py:72-78)
        self.postnet_dim = int(mc.get("_postnet_dim", 512))
        self.postnet_kernel = 5
        self.postnet_layers = 5
        self.postnet_dropout = 0.5
        self.model_config = mc
        self.preprocess_config = pc
