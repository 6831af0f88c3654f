#!/usr/bin/env python3
"""Diagnostic: smoke()'s MAML parity numbers under the current MTTS_* environment, for inner lr 1e-3 and 1e-4."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from meta_tts_amd import synth
from meta_tts_amd.config import ModelDims, default_algorithm_config
from meta_tts_amd.engine import Engine
from oracle import fs2_oracle as O
dims = ModelDims(); mods = default_algorithm_config()["adapt"]["modules"]
small = dict(s_range=(6, 13), d_range=(1, 7), first_len=12)
sup = synth.make_batch(21, 3, speaker=9, **small); qry = synth.make_batch(22, 3, speaker=9, **small)
params = synth.make_params(dims, 0)
eng = Engine(dims, adapt_modules=mods, max_tasks=1, max_B=3, max_S=16, max_T=96)
eng.load_params(params); eng.set_batches(0, [sup]); eng.set_batches(1, [qry], spk_from=[sup], average_spk=True)
p = {k: torch.from_numpy(v.copy()) for k, v in params.items()}
for k in p:
    if not k.endswith(("position_enc", "pitch_bins", "energy_bins")): p[k].requires_grad_(True)
buf = {k: torch.from_numpy(v.copy()) for k, v in synth.make_buffers(dims).items()}
tb = O.to_torch_batch(sup)
for lr in (1e-3, 1e-4):
    q, s = eng.meta_grad(5, lr, 1.0)
    ql, sl, _, _ = O.maml_task(p, buf, tb, O.to_torch_batch(qry), steps=5, lr=lr, second_order=False, modules=mods, n_head=(dims.enc_heads, dims.dec_heads))
    out = []
    for name in ("mel_linear.weight", "decoder.layer_stack.5.pos_ffn.w_2.weight"):
        g = torch.autograd.grad(ql[0], p[name], retain_graph=True)[0].numpy(); got = eng.export(name, 1)
        out.append(float(np.abs(got - g).max() / max(np.abs(g).max(), 1e-8)))
    print(f"lr {lr}: sup losses dev {s[:, 0, 0].round(4).tolist()} oracle {[round(float(x[0]), 4) for x in sl]} | q {q[0,0]:.5f} vs {float(ql[0]):.5f} | grad rel {out}")
