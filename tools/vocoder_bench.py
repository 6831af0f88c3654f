#!/usr/bin/env python3
"""MelGAN generator alone on 5 utterances (1881 mel frames), device-to-device entry; ms per call.  GPU only."""
import sys, numpy as np, math
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from meta_tts_amd import vocoder as V
voc = V.MelGAN(max_B=5, max_T=600)
voc.set_stream(torch.cuda.current_stream().cuda_stream)
g = np.random.RandomState(0)
lens = np.array([420, 380, 350, 390, 341], np.int32)
x = torch.from_numpy((g.standard_normal((5, 420, 80)) - 4).astype(np.float32)).cuda()
wav = torch.zeros(5, 420 * 256, device="cuda")
for it in range(6):
    if it == 1:
        torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
    voc.mel2wav_device(x.data_ptr(), 0, 5, 420, lens, wav.data_ptr(), 1 / math.log(10))
e1.record(); torch.cuda.synchronize()
print("vocoder ms per call (1881 frames):", e0.elapsed_time(e1) / 5)
