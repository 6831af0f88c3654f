#!/usr/bin/env python3
"""BASELINE config C2 alone (bench.py's baseline_c2 leg: fp32 and bf16 numerics, per-mode GEMM roofline) — one JSON object."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from meta_tts_amd.config import ModelDims, default_train_config
torch.cuda.set_device(0)
print(json.dumps(bench.baseline_c2_leg(ModelDims(), 0, bench.noam_lr, default_train_config()["optimizer"], iters=int(os.environ.get("C2_ITERS", "10")),
                                   modes=tuple(os.environ.get("C2_MODES", "fp32,bf16").split(",")))))
