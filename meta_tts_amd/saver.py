"""Result-directory layout of the test / validation stage — what the reference's `evaluation/` scripts consume
(lightning/callbacks/saver.py:130-178 `on_test_batch_end`, :203-213 `log_csv`, :86-90 validation CSVs; callbacks/utils.py:55-141
`recon_samples` / `synth_samples`; evaluation/compute_mos.py:69-90 and wavs_to_dvector.py:247-294 glob these paths):

    <result_dir>/csv/Testing/step_<global_step>/<task_id>.csv                       Step,Total Loss,...,Duration Loss per ft_step
    <result_dir>/audio/Testing/step_<global_step>/<task_id>/<basename>.recon.wav    vocoder(ground-truth mel), written at ft_step 0
    <result_dir>/audio/Testing/step_<global_step>/<task_id>/<basename>.step_<global_step>-FTstep_<ft>.synth.wav
    <result_dir>/figure/Testing/step_<global_step>/<task_id>/                       (directory only: matplotlib plots are not produced)
    <log_dir>/csv/Validation/<task_id>.csv                                          one appended row per validation pass

`task_id` = `test_SQids2Tid["-".join(sup_ids) + "." + "-".join(qry_ids)]` (datamodules/utils.py:96-105; meta_tts_amd.data.prefetch_tasks),
suffixed `_<i>` in the 1-shot mode where one task yields several outputs.  CSV bytes are what pandas' `DataFrame.to_csv` writes
for the same values (shortest round-trip float repr), without importing pandas."""
from __future__ import annotations

import os
from typing import Dict, Optional, Sequence

import numpy as np

from .engine import LOSS_NAMES

CSV_COLUMNS = list(LOSS_NAMES)  # saver.py:19


def _fmt(v) -> str:
    """pandas' to_csv float formatting (repr of the Python float: shortest string that round-trips)."""
    return repr(float(v))


def loss2dict(loss) -> Dict[str, float]:
    """lightning/utils.py:65-74 (values as Python floats)."""
    return {k: float(v) for k, v in zip(CSV_COLUMNS, loss)}


class Saver:
    """The file-writing half of the reference's `Saver` callback (Comet / TensorBoard logging and figure plotting are not on
    the hot path)."""

    def __init__(self, preprocess_config, log_dir: str, result_dir: str):
        self.preprocess_config = preprocess_config
        self.log_dir, self.result_dir = log_dir, result_dir
        os.makedirs(log_dir, exist_ok=True)
        os.makedirs(result_dir, exist_ok=True)

    # saver.py:203-213
    def log_csv(self, stage: str, step: int, basename: str, loss_dict: Dict[str, float]) -> str:
        root = self.log_dir if stage in ("Training", "Validation") else self.result_dir
        d = os.path.join(root, "csv", stage)
        os.makedirs(d, exist_ok=True)
        path = os.path.join(d, f"{basename}.csv")
        new = not os.path.exists(path)
        with open(path, "a", newline="") as f:
            if new:
                f.write(",".join(["Step"] + CSV_COLUMNS) + "\n")
            f.write(",".join([str(int(step))] + [_fmt(loss_dict[c]) for c in CSV_COLUMNS]) + "\n")
        return path

    # saver.py:76-90 (the per-task CSV row; figure / audio logging goes to the experiment logger in the reference)
    def on_validation_batch_end(self, outputs, batch, global_step: int, val_SQids2Tid: Dict[str, str]) -> str:
        sup_ids, qry_ids = batch[0][0][0][0], batch[0][1][0][0]
        task_id = val_SQids2Tid[f"{'-'.join(sup_ids)}.{'-'.join(qry_ids)}"]
        return self.log_csv("Validation", global_step + 1, task_id, loss2dict(outputs["losses"]))

    # saver.py:130-178
    def on_test_batch_end(self, all_outputs: Sequence[dict], batch, test_SQids2Tid: Dict[str, str], global_step: int, adaptation_steps: int,
                          test_adaptation_steps: int, vocoder=None):
        sup_ids, qry_ids = batch[0][0][0][0], batch[0][1][0][0]
        _task_id = test_SQids2Tid[f"{'-'.join(sup_ids)}.{'-'.join(qry_ids)}"]
        written = []
        for i, outputs in enumerate(all_outputs):
            task_id = _task_id if len(all_outputs) == 1 else f"{_task_id}_{i}"
            figure_dir = os.path.join(self.result_dir, "figure", "Testing", f"step_{global_step}", task_id)
            audio_dir = os.path.join(self.result_dir, "audio", "Testing", f"step_{global_step}", task_id)
            log_dir = os.path.join(self.result_dir, "csv", "Testing", f"step_{global_step}")
            for d in (figure_dir, audio_dir, log_dir):
                os.makedirs(d, exist_ok=True)
            csv_path = os.path.join(log_dir, f"{task_id}.csv")
            rows = []
            _batch = outputs["_batch"]
            for ft_step in range(0, test_adaptation_steps + 1, adaptation_steps):
                o = outputs[f"step_{ft_step}"]
                if ft_step == 0 and vocoder is not None:
                    self.recon_samples(_batch, o["recon"]["output"], vocoder, audio_dir)
                if "recon" in o:
                    rows.append((ft_step, loss2dict(o["recon"]["losses"])))
                if "synth" in o and vocoder is not None:
                    self.synth_samples(_batch, o["synth"]["output"], vocoder, audio_dir, f"step_{global_step}-FTstep_{ft_step}")
            with open(csv_path, "a", newline="") as f:   # mode='a', header=True: a re-run appends a second block, as the reference does
                f.write(",".join(["Step"] + CSV_COLUMNS) + "\n")
                for ft_step, ld in rows:
                    f.write(",".join([str(ft_step)] + [_fmt(ld[c]) for c in CSV_COLUMNS]) + "\n")
            written.append(csv_path)
        return written

    # ---- audio (callbacks/utils.py:55-141) ------------------------------------------------------------------------------------
    def _hop_and_rate(self):
        p = self.preprocess_config["preprocessing"]
        return int(p["stft"]["hop_length"]), int(p["audio"]["sampling_rate"]), float(p["audio"]["max_wav_value"])

    def _write(self, path: str, rate: int, wav: np.ndarray):
        from scipy.io import wavfile
        wavfile.write(path, rate, np.asarray(wav, np.int16))

    def recon_samples(self, targets, predictions, vocoder, audio_dir: str):
        """`<basename>.recon.wav`: the vocoder on the GROUND-TRUTH mels, cropped to mel_len * hop samples."""
        hop, rate, max_wav = self._hop_and_rate()
        mels = np.asarray(targets[6], np.float32).transpose(0, 2, 1)
        lengths = [int(l) * hop for l in np.asarray(predictions[9])]
        for wav, basename in zip(vocoder.infer(mels, max_wav, lengths=lengths), targets[0]):
            self._write(os.path.join(audio_dir, f"{basename}.recon.wav"), rate, wav)

    def synth_samples(self, targets, predictions, vocoder, audio_dir: str, name: str):
        """`<basename>.<name>.synth.wav`: the vocoder on the predicted post-net mels."""
        hop, rate, max_wav = self._hop_and_rate()
        mels = np.asarray(predictions[1], np.float32).transpose(0, 2, 1)
        lengths = [int(l) * hop for l in np.asarray(predictions[9])]
        for wav, basename in zip(vocoder.infer(mels, max_wav, lengths=lengths), targets[0]):
            self._write(os.path.join(audio_dir, f"{basename}.{name}.synth.wav"), rate, wav)
