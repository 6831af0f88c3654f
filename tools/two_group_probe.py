#!/usr/bin/env python3
"""Would the 8-task meta-step be faster as TWO groups of 4 tasks on two streams (two handles), each group's launch tails filled by the other group's work?
Times per meta-gradient: one handle with 8 tasks; two handles with 4 tasks each, enqueued back to back on two streams; (control) the two 4-task handles one
after the other on one stream.  First order, dropout on, no optimizer step (the gradient call is > 99 % of the step)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from meta_tts_amd import synth
from meta_tts_amd.config import ModelDims, default_algorithm_config
from meta_tts_amd.engine import Engine

dims, mods = ModelDims(), default_algorithm_config()["adapt"]["modules"]
tasks = [synth.make_task(j) for j in range(8)]
max_T = max(max(s[8], q[8]) for s, q in tasks)
params = synth.make_params(dims, 0, weight_scale=0.5)


def make(ts, stream):
    e = Engine(dims, adapt_modules=mods, max_tasks=len(ts), max_B=5, max_S=80, max_T=max_T)
    e.set_stream(stream.cuda_stream)
    e.load_params(params)
    e.set_dropout(True, 1234)
    e.set_batches(0, [t[0] for t in ts]); e.set_batches(1, [t[1] for t in ts], spk_from=[t[0] for t in ts], average_spk=True)
    return e


def timeit(fn, n=8):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
order = 2 if "--so" in sys.argv else 1
e8 = make(tasks, s0)
t8 = timeit(lambda: e8.meta_grad(5, 1e-3, 0.125, second_order=(order == 2), fetch_losses=False))
e8.close()
ea, eb = make(tasks[:4], s0), make(tasks[4:], s1)
def both():
    ea.meta_grad(5, 1e-3, 0.125, second_order=(order == 2), fetch_losses=False)
    eb.meta_grad(5, 1e-3, 0.125, second_order=(order == 2), fetch_losses=False)
t44 = timeit(both)
eb.set_stream(s0.cuda_stream)
t44s = timeit(both)
print(f"order {order}: one handle x 8 tasks {t8:.2f} ms | two handles x 4 tasks on two streams {t44:.2f} ms | the same two on ONE stream {t44s:.2f} ms")
