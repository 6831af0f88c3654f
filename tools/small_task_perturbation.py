#!/usr/bin/env python3
"""CPU, oracle only: the SMALL MAML task of the reference fixtures (tests/golden/maml_small_lr1e-3 / lr2e-3.npz) on UNSCALED random weights — the first-order query
gradient with and without a 1e-7 relative random perturbation of every weight, six seeds: how discontinuous is the function the loose fixtures pin?  (profiles/r06_unscaled_maml.md)"""
import os, sys, numpy as np, torch
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests'))
from oracle_util import O, SMALL, heads, synth, torch_buffers, torch_params
from meta_tts_amd.config import ModelDims, default_algorithm_config
torch.set_num_threads(4)
DIMS=ModelDims(); MODS=default_algorithm_config()["adapt"]["modules"]
sup,qry=synth.make_batch(21,3,speaker=9,**SMALL), synth.make_batch(22,3,speaker=9,**SMALL)
names=["mel_linear.weight","decoder.layer_stack.5.pos_ffn.w_2.weight","decoder.layer_stack.0.slf_attn.w_qs.weight","postnet.convolutions.2.0.conv.weight","variance_adaptor.pitch_embedding.weight"]
def run(lr, eps, seed):
    p=torch_params(DIMS,requires_grad=False,weight_scale=1.0)
    g=torch.Generator().manual_seed(seed)
    for k in p:
        if not k.endswith(("position_enc","pitch_bins","energy_bins")):
            if eps: p[k]=p[k]*(1+eps*torch.randn(p[k].shape,generator=g))
            p[k].requires_grad_(True)
    ql,sl,_,_=O.maml_task(p,torch_buffers(DIMS),O.to_torch_batch(sup),O.to_torch_batch(qry),steps=5,lr=lr,second_order=False,modules=MODS,n_head=heads(DIMS))
    gs=torch.autograd.grad(ql[0],[p[n] for n in names])
    return [x.numpy() for x in gs], [float(l[0]) for l in sl]
for lr in (1e-3, 2e-3):
    base, sl = run(lr, 0.0, 0)
    print("lr", lr, "support losses", [round(x,3) for x in sl])
    for seed in range(1,7):
        g,_ = run(lr, 1e-7, seed)
        print("  seed", seed, " ".join(f"{np.abs(a-b).max()/np.abs(a).max():.1e}" for a,b in zip(base,g)))
