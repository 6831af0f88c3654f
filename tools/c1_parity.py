#!/usr/bin/env python3
"""mel L1 of the HIP path vs the reference fixture (C1, S=80, T=555) for the current MTTS_NUMERICS mode. GPU only."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from meta_tts_amd import synth
from meta_tts_amd.config import ModelDims
from meta_tts_amd.engine import Engine
g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "c1_forward.npz"))
d = ModelDims(); eng = Engine(d, adapt_modules=[], max_tasks=1, max_B=1, max_S=80, max_T=555); eng.load_params(synth.make_params(d, 0))
eng.set_batches(0, [synth.make_batch(0, 1)])
for tr, key in ((False, "mel_post"), (True, "train_mel_post")):
    eng.forward(0, train=tr); o = eng.outputs(0, 0)["mel_post"]; dd = np.abs(o - g[key])
    print(f"numerics={os.environ.get('MTTS_NUMERICS','0')} train={tr}: mel L1 {dd.mean():.3e} max {dd.max():.3e}")
