"""Diagnostic (CPU, oracle only): ReLU pre-activations of one task's QUERY pass that lie inside fp32 noise of zero, with the pitch buckets of the
positions a predictor kink reaches (profiles/r05_dropout_parity.md).  Usage: python tools/relu_kink_probe.py"""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from oracle_util import O, heads, synth, torch_buffers, torch_params
from oracle.dropout_masks import DropoutMasks, plan_seed
from meta_tts_amd.config import ModelDims, default_algorithm_config
import torch.nn.functional as F
torch.set_num_threads(8)
DIMS=ModelDims(); MODS=default_algorithm_config()["adapt"]["modules"]
j=3; seed=1234
sup,qry=synth.make_task(j)
p=torch_params(DIMS,requires_grad=True,weight_scale=0.5); buf=torch_buffers(DIMS)
dms=[DropoutMasks(plan_seed(seed,k+1),j) for k in range(6)]
rec=[]; orig=F.relu; state={"on":False}
def relu(x,*a,**k):
    if state["on"]: rec.append(x.detach())
    return orig(x,*a,**k)
O.F.relu=relu
# run inner steps w/o recording, then query with recording: replicate maml_task but flag the query pass
names=O.adapted_names(p,MODS); fast={k:p[k] for k in names}
tb_s,tb_q=O.to_torch_batch(sup),O.to_torch_batch(qry)
for s in range(5):
    cur=dict(p); cur.update(fast)
    preds=O.fs2_forward(cur,buf,*tb_s[2:],n_head=heads(DIMS),training=True,dropout=dms[s])
    loss=O.fs2_loss(tb_s,preds)
    g=torch.autograd.grad(loss[0],[fast[k] for k in names])
    fast={k:fast[k]-0.001*gi for k,gi in zip(names,g)}
state["on"]=True
cur=dict(p); cur.update(fast)
preds=O.fs2_forward(cur,buf,tb_s[2],*tb_q[3:],n_head=heads(DIMS),training=True,average_spk_emb=True,dropout=dms[5])
state["on"]=False
pidx=torch.bucketize(tb_q[9], p["variance_adaptor.pitch_bins"])
print("relu calls",len(rec))
for i,x in enumerate(rec):
    a=x.abs()
    m=a.min()
    n_small=int((a<3e-7).sum())
    if n_small:
        idx=(a<3e-7).nonzero()
        print(i, tuple(x.shape), "min|pre|",float(m), "count<3e-7",n_small, idx[:5].tolist())
        if x.dim()==3 and x.shape[1]==256 and x.shape[2]==pidx.shape[1]:
            for b,c,s in idx[:5].tolist():
                print("    buckets around:", [int(pidx[b,t]) for t in range(max(0,s-1),min(pidx.shape[1],s+2))], "src_len", int(tb_q[4][b]), "s", s)
