#!/usr/bin/env python3
"""Host time of the calls of one meta-step, from an idle stream (no device sync between them): ingestion, the gradient call, the outer update — and
the device time of the whole step.  Usage: host_enqueue_probe.py [tasks_per_rank]   (8 = the 8-task step, 1 = the single-task rank)"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B
from meta_tts_amd import synth
from meta_tts_amd.config import ModelDims, default_algorithm_config
from meta_tts_amd.engine import Engine

nt = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dims = ModelDims()
tasks = [synth.make_task(j) for j in range(nt)]
max_T = max(max(s[8], q[8]) for s, q in tasks)
eng = Engine(dims, adapt_modules=default_algorithm_config()["adapt"]["modules"], max_tasks=nt, max_B=5, max_S=80, max_T=max_T, device=0)
eng.set_stream(torch.cuda.current_stream().cuda_stream)
eng.load_params(synth.make_params(dims, 0, weight_scale=B.WEIGHT_SCALE))
eng.set_dropout(True, 1234)
sup, qry = [t[0] for t in tasks], [t[1] for t in tasks]
def step(rec=None):
    t = [time.perf_counter()]
    eng.set_batches(0, sup); eng.set_batches(1, qry, spk_from=sup, average_spk=True); t.append(time.perf_counter())
    eng.meta_grad(B.INNER_STEPS, B.INNER_LR, 1.0 / 8, second_order=False, fetch_losses=False); t.append(time.perf_counter())
    eng.outer_update(lr=1e-4, betas=(0.9, 0.98), eps=1e-9, weight_decay=0.0, max_norm=1.0); t.append(time.perf_counter())
    if rec is not None: rec.append(np.diff(t) * 1e3)
for _ in range(3): step()
torch.cuda.synchronize()
rec = []
for _ in range(5):
    t0 = time.perf_counter(); step(rec); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    rec[-1] = np.append(rec[-1], [(t1 - t0) * 1e3, (t2 - t0) * 1e3])
r = np.median(np.array(rec), axis=0)
print(f"tasks/rank {nt}: host ms — ingest {r[0]:.2f}, meta_grad {r[1]:.2f}, outer_update {r[2]:.2f}, all calls {r[3]:.2f}; device done at {r[4]:.2f}")
