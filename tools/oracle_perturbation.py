"""Diagnostic (CPU, oracle only): is the 5-step inner loop chaotic?  One task's first-order query gradient with and without a 1e-7 relative random
perturbation of every weight, dropout on / off (profiles/r05_dropout_parity.md).  Usage: python tools/oracle_perturbation.py 0.5,0.35"""
import os, sys, numpy as np, torch, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from oracle_util import O, heads, synth, torch_buffers, torch_params
from oracle.dropout_masks import DropoutMasks, plan_seed
from meta_tts_amd.config import ModelDims, default_algorithm_config
torch.set_num_threads(8)
DIMS=ModelDims(); MODS=default_algorithm_config()["adapt"]["modules"]
j=3; seed=1234
sup,qry=synth.make_task(j)
names=["postnet.convolutions.4.0.conv.weight","postnet.convolutions.2.0.conv.weight","variance_adaptor.energy_predictor.conv_layer.conv1d_1.conv.weight","mel_linear.weight","decoder.layer_stack.5.pos_ffn.w_2.weight","decoder.layer_stack.0.slf_attn.w_qs.weight","variance_adaptor.pitch_embedding.weight"]
for ws in [float(x) for x in sys.argv[1].split(",")]:
  for drop in (1,0):
    res=[]
    for eps in (0.0, 1e-7):
        p=torch_params(DIMS,requires_grad=False,weight_scale=ws)
        g=torch.Generator().manual_seed(0)
        for k in p:
            if not k.endswith(("position_enc","pitch_bins","energy_bins")):
                if eps: p[k]=p[k]*(1+eps*torch.randn(p[k].shape,generator=g))
                p[k].requires_grad_(True)
        dms=[DropoutMasks(plan_seed(seed,k+1),j) for k in range(6)] if drop else None
        t0=time.time()
        ql,sl,_,_=O.maml_task(p,torch_buffers(DIMS),O.to_torch_batch(sup),O.to_torch_batch(qry),steps=5,lr=1e-3,second_order=False,modules=MODS,n_head=heads(DIMS),dropout=dms)
        gs=torch.autograd.grad(ql[0],[p[n] for n in names])
        res.append((float(ql[0]),[float(l[0]) for l in sl],[x.numpy() for x in gs]))
    a,b=res
    print(f"ws {ws} dropout {drop}: support loss per step {[round(x,4) for x in a[1]]} query {a[0]:.5f}; loss diff {abs(a[0]-b[0])/abs(a[0]):.1e}; ({time.time()-t0:.0f}s/run)")
    for n,x,y in zip(names,a[2],b[2]):
        print(f"   {np.abs(x-y).max()/np.abs(x).max():.2e}  {n}")
