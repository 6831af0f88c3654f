#!/usr/bin/env python3
"""on UNSCALED random weights five inner steps at lr 1e-3 / 2e-3 are expansive and the reference fixtures only bound the engine loosely
(rtol 1e-1 / 2e-1).  How far is each fp32 party — the engine, the fp32 oracle — from a FLOAT64 evaluation of the same small task?  (oracle/arbiter.py, first order)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from oracle_util import O, SMALL, heads, synth, torch_buffers, torch_params
from oracle import arbiter as ARB
from meta_tts_amd.config import ModelDims, default_algorithm_config
from meta_tts_amd.engine import Engine
torch.set_num_threads(16)
DIMS, MODS = ModelDims(), default_algorithm_config()["adapt"]["modules"]
NAMES = ["mel_linear.weight", "decoder.layer_stack.5.pos_ffn.w_2.weight", "decoder.layer_stack.0.slf_attn.w_qs.weight", "postnet.convolutions.2.0.conv.weight",
         "variance_adaptor.duration_predictor.linear_layer.weight", "variance_adaptor.pitch_embedding.weight", "postnet.convolutions.4.1.weight",
         "decoder.layer_stack.2.slf_attn.layer_norm.weight", "encoder.layer_stack.0.slf_attn.w_qs.bias"]
sup, qry = synth.make_batch(21, 3, speaker=9, **SMALL), synth.make_batch(22, 3, speaker=9, **SMALL)
for lr, scale in ((1e-4, 1.0), (1e-3, 1.0), (2e-3, 1.0), (1e-3, 0.5)):
    eng = Engine(DIMS, adapt_modules=MODS, max_tasks=1, max_B=3, max_S=16, max_T=96)
    eng.load_params(synth.make_params(DIMS, 0, weight_scale=scale))
    eng.set_batches(0, [sup]); eng.set_batches(1, [qry], spk_from=[sup], average_spk=True)
    eng.meta_grad(5, lr, 1.0)
    out = eng.outputs(1, 0)
    engine = {"grads": {n: eng.export(n, 1) for n in NAMES}, "mel": out["mel"], "mel_post": out["mel_post"]}
    eng.close()
    p = torch_params(DIMS, requires_grad=True, weight_scale=scale)
    ql, _, _, qp = O.maml_task(p, torch_buffers(DIMS), O.to_torch_batch(sup), O.to_torch_batch(qry), steps=5, lr=lr, second_order=False, modules=MODS, n_head=heads(DIMS))
    gs = torch.autograd.grad(ql[0], [p[n] for n in NAMES], allow_unused=True)
    o32 = {"grads": {n: (g.numpy() if g is not None else np.zeros(tuple(p[n].shape), np.float32)) for n, g in zip(NAMES, gs)}, "mel": qp[0].detach().numpy(), "mel_post": qp[1].detach().numpy()}
    rep = ARB.arbitrate_task(synth.make_params(DIMS, 0, weight_scale=scale), synth.make_buffers(DIMS), sup, qry, modules=MODS, n_head=heads(DIMS), max_seq_len=DIMS.max_seq_len,
                             steps=5, lr=lr, masks=None, names=NAMES, parties={"engine": engine, "oracle32": o32}, explain=False)
    e = [rep["tensors"][n]["engine"]["l1"] for n in NAMES]; o = [rep["tensors"][n]["oracle32"]["l1"] for n in NAMES]
    d = [float(np.abs(engine["grads"][n] - o32["grads"][n]).max() / max(np.abs(o32["grads"][n]).max(), 1e-30)) for n in NAMES]
    print(f"lr {lr:g} weights x{scale}: engine vs f64 max {max(e):.2e} median {np.median(e):.2e} | oracle32 vs f64 max {max(o):.2e} median {np.median(o):.2e} | engine vs oracle32 max {max(d):.2e} | gate pass {rep['pass']}")
