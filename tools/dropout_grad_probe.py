"""Diagnostic (GPU): every parameter gradient of the 8-task grouped first-order meta-gradient with dropout on, per task, against
the oracle with the engine's masks.  Usage: python tools/dropout_grad_probe.py [tasks to check, e.g. 3,0] [dropout 0/1]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle_util import O, heads, synth, torch_buffers, torch_params  # noqa: E402
from oracle.dropout_masks import DropoutMasks, plan_seed  # noqa: E402
from meta_tts_amd.config import ModelDims, default_algorithm_config  # noqa: E402
from meta_tts_amd.engine import Engine  # noqa: E402

torch.set_num_threads(16)
DIMS = ModelDims()
MODS = default_algorithm_config()["adapt"]["modules"]
check = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "3").split(",")]
drop = int(sys.argv[2]) if len(sys.argv) > 2 else 1
STEPS = int(os.environ.get("PROBE_STEPS", "5"))
WS = float(os.environ.get("PROBE_WEIGHT_SCALE", "0.5"))
seed = 1234
tasks = [synth.make_task(j) for j in range(8)]
max_T = max(max(s[8], q[8]) for s, q in tasks)
eng = Engine(DIMS, adapt_modules=MODS, max_tasks=8, max_B=5, max_S=80, max_T=max_T)
eng.load_params(synth.make_params(DIMS, 0, weight_scale=WS))
sup, qry = [t[0] for t in tasks], [t[1] for t in tasks]
runs = []
for rep in range(2):
    eng.set_batches(0, sup)
    eng.set_batches(1, qry, spk_from=sup, average_spk=True)
    eng.set_dropout(bool(drop), seed)
    q, s = eng.meta_grad(STEPS, 0.001, 1.0 / 8)
    runs.append({j: {n: eng.export(n, 2, j) * 8.0 for n in eng.params} for j in check})
    qouts = {j: eng.outputs(1, j) for j in check}
for j in check:
    d = max(float(np.abs(runs[0][j][n] - runs[1][j][n]).max()) for n in eng.params)
    print(f"task {j}: run-to-run max abs diff over all tensors {d:.3e}")
p = torch_params(DIMS, requires_grad=True, weight_scale=WS)
buf = torch_buffers(DIMS)
for j in check:
    dms = [DropoutMasks(plan_seed(seed, k + 1), j) for k in range(STEPS + 1)] if drop else None
    ql, sl, _, qpreds = O.maml_task(p, buf, O.to_torch_batch(sup[j]), O.to_torch_batch(qry[j]), steps=STEPS, lr=0.001, second_order=False, modules=MODS,
                               n_head=heads(DIMS), dropout=dms)
    print(f"steps {STEPS} weight_scale {WS} dropout {drop} | task {j}: query loss rel err {abs(q[j, 0] - float(ql[0])) / abs(float(ql[0])):.2e}")
    for key, ref in (("mel", qpreds[0]), ("mel_post", qpreds[1]), ("p", qpreds[2]), ("e", qpreds[3]), ("logd", qpreds[4])):
        a, b = qouts[j][key], ref.detach().numpy()
        dd = np.abs(a - b[:, :a.shape[1]] if a.ndim > 1 else a - b)
        print(f"   query-pass output {key}: max|diff| {dd.max():.3e} (max|ref| {np.abs(b).max():.3e}), elements with |diff| > 1e-4: {int((dd > 1e-4).sum())}")
    tgt = qry[j][6]
    for key, ref in (("mel", qpreds[0]), ("mel_post", qpreds[1])):
        a, b = qouts[j][key], ref.detach().numpy()
        T = a.shape[1]
        sa, sb = np.sign(a - tgt[:, :T]), np.sign(b[:, :T] - tgt[:, :T])
        valid = np.arange(T)[None, :] < np.asarray(qry[j][7])[:, None]
        print(f"   L1 sign flips in {key} (valid frames): {int(((sa != sb) & valid[:, :, None]).sum())} of {int(valid.sum()) * a.shape[2]}")
    names = [n for n in eng.params if n.split('.')[0] in MODS]
    gs = torch.autograd.grad(ql[0], [p[n] for n in names], allow_unused=True)
    rows = []
    for n, g in zip(names, gs):
        ref = g.numpy()
        got = runs[0][j][n]
        rows.append((float(np.abs(got - ref).max() / max(float(np.abs(ref).max()), 1e-30)), n, float(np.abs(ref).max())))
    gmax = max(m for _, _, m in rows)
    rows = [x for x in rows if x[2] > 1e-5 * gmax]       # (tensors whose gradient is zero in exact arithmetic are pure rounding noise)
    rows.sort(reverse=True)
    for r, n, m in rows[:8]:
        print(f"   {r:.3e}  {n}  (max |ref| {m:.3e})")
    for r, n2, m in sorted(rows, key=lambda x: x[1]):
        if n2.startswith("postnet.") and n2.endswith(("conv.weight", "1.weight")):
            print(f"   [postnet] {r:.3e}  {n2}")
    n = "variance_adaptor.pitch_embedding.weight"
    ref = dict(zip(names, gs))[n].numpy(); got = runs[0][j][n]
    bad = np.argsort(-np.abs(got - ref).max(axis=1))[:4]
    for r_ in bad:
        print(f"   pitch_emb row {r_}: max|diff| {np.abs(got[r_] - ref[r_]).max():.3e} max|ref| {np.abs(ref[r_]).max():.3e}  bucket of 0.0 = {int(np.searchsorted(p['variance_adaptor.pitch_bins'].numpy(), 0.0))}")
eng.close()
