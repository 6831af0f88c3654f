#!/usr/bin/env python3
"""bench.py — meta-steps/sec of the Meta-TTS hot path on MI355X (BASELINE.json metric).

One "step" = one meta-step of config C3 (BASELINE.json configs[2]):
8 tasks x [5 inner SGD steps on 5 support utterances + query pass on 5 query utterances]
(first-order MAML, fp32), outer-gradient mean over the 8 tasks, clip_grad_norm_(1.0), Adam with the
Noam schedule.  Synthetic LibriTTS-shaped batches (meta_tts_amd.synth), random-init weights, inputs
resident in HBM before the timed region.  With --gpus N the 8 tasks are sharded 8/N per rank
(one process per GPU, launched by torch.distributed.run) and the flat outer gradient is
all-reduced over RCCL/xGMI; the meta-batch stays 8 tasks, so scaling is "strong".

Extra legs (rank 0, N=1 only): `roofline` times every launch of the grouped fp32-MFMA GEMM family
with HIP events on the launch stream (mtts_profile_gemm) over one more meta-step and reports the
dominant instantiation against the 157.3 TFLOP/s fp32-matrix peak; `cpu_baseline` times the oracle
(torch fp32 restatement, kind "port") on the host cores for a bounded sample of the same workload.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

META_BATCH = 8
INNER_STEPS = 5
INNER_LR = 0.001
FP32_MATRIX_PEAK_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md, chip-level parameters
PARITY_RTOL = 2e-3  # per-task query losses of the timed configuration (dropout off) vs the oracle; the line is refused above it
# Sampled per-task query-gradient tensors are judged by the float64 arbiter (oracle/arbiter.py): err(engine, fp64) <= 3 * err(oracle32, fp64) + 1e-3
# per tensor (max-norm relative to the tensor's largest fp64 entry), L1 signs of the elements inside fp32 noise of their target taken from each
# party's own forward output, single ReLU units inside fp32 noise of zero priced one by one where a tensor still fails.  No L2 escape.
# Random-init weights of the timed meta-step: every Linear / Conv1d weight matrix x 0.5 (synth.make_params), the initialisation on which five
# inner SGD steps at the reference's lr 1e-3 are contractive (the support loss falls), as in tests/golden/maml_small_lr1e-3_scaled.npz and
# tests/test_gpu_timed_config.py.  On unscaled random weights the inner loop is expansive: summation-order differences between any two
# correct fp32 implementations are amplified to several per cent of the query gradient after 5 steps (measured 3.6-5.5 %), which makes a
# gradient comparison meaningless.  Timing does not depend on the weight values.
WEIGHT_SCALE = 0.5
EXCHANGE_RTOL = 5e-3   # N > 1: all-reduced outer gradient vs the single-handle 8-task gradient (grouped-vs-alone bound of tests/test_gpu_timed_config.py)


def noam_lr(step, d_model=256, warm=4000, anneal=(300000, 400000, 500000), rate=0.3):
    cur = step + 1
    lr = min(cur ** -0.5, warm ** -1.5 * cur)
    for s in anneal:
        if cur > s:
            lr *= rate
    return d_model ** -0.5 * lr


def spawn_command(n_gpus, argv, port=None):
    """The command `bench.py --gpus N` re-executes itself under when it was started as ONE process: one rank per GPU through
    torch.distributed.run, rendezvous on 127.0.0.1 (the reference's launcher, pl.Trainer(strategy="ddp"), main.py:30-38, also
    spawns its own ranks)."""
    if port is None:
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def maybe_self_launch(args):
    """--gpus N > 1 without a launcher (WORLD_SIZE unset): spawn the N ranks and relay rank 0's JSON line.  Returns the child's
    exit code, or None when this process is itself a rank (or N == 1)."""
    if args.gpus <= 1 or "WORLD_SIZE" in os.environ:
        return None
    import subprocess
    env = dict(os.environ)
    if getattr(args, "selftest_emu", False):
        env.setdefault("OMP_NUM_THREADS", "2")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC for RCCL between the ranks
    return subprocess.call(spawn_command(args.gpus, sys.argv[1:]), env=env)


def spawn_selftest(args):
    """CPU-only check of the launch plumbing (tests/test_bench_spawn.py): every rank joins a gloo group, one all_reduce, rank 0
    prints a line whose n_gpus is the group's real size.  No engine, no timing — not a bench result."""
    import torch
    import torch.distributed as dist
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"refusing to report: {world} rank(s) running but --gpus {args.gpus}")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    t = torch.ones(4) * (rank + 1)
    if world > 1:
        dist.all_reduce(t)
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"selftest": "spawn", "n_gpus": world, "allreduce_sum": float(t[0]), "tasks_per_rank": META_BATCH // world}))


# outer-gradient tensors compared between the timed configuration and the oracle (bench.py parity_check): one per module family
GRAD_SAMPLES = ("mel_linear.weight", "decoder.layer_stack.5.pos_ffn.w_2.weight", "decoder.layer_stack.0.slf_attn.w_qs.weight",
                "encoder.layer_stack.3.pos_ffn.w_1.weight", "postnet.convolutions.2.0.conv.weight",
                "variance_adaptor.pitch_embedding.weight", "speaker_emb.model.weight")


# allocator / OpenMP-runtime settings of the concurrent CPU-baseline workers (set in the worker before torch loads; see cpu_baseline)
CPU_RUNTIME_ENV = {"OMP_WAIT_POLICY": "PASSIVE", "GOMP_SPINCOUNT": "0", "MALLOC_ARENA_MAX": "1", "MALLOC_MMAP_THRESHOLD_": "33554432",
                   "MALLOC_TRIM_THRESHOLD_": "4294967295", "MALLOC_TOP_PAD_": "268435456"}
DROPOUT_SEED = 1234      # mtts_set_dropout(h, 1, DROPOUT_SEED + rank) for the timed steps AND for the parity run


def _oracle_masks(dims, sup, qry, task, seed=DROPOUT_SEED, steps=None):
    """The engine's dropout masks of one task's 5 inner steps + query pass (plan seeds in draw order), generated BEFORE any clock
    starts: the timed oracle then pays one multiply per dropout site, as the reference does."""
    from oracle.dropout_masks import DropoutMasks, plan_seed
    probs = dict(enc=dims.enc_dropout, dec=dims.dec_dropout, vp=dims.vp_dropout, postnet=0.5)
    out = []
    steps = INNER_STEPS if steps is None else steps
    for k in range(steps + 1):
        b = sup if k < steps else qry
        out.append(DropoutMasks(plan_seed(seed, k + 1), task, probs).precompute(
            len(b[4]), int(b[5]), [int(x) for x in b[7]], int(b[8]), enc_layers=dims.enc_layers, dec_layers=dims.dec_layers, d_model=dims.d_model,
            vp_filter=dims.vp_filter, postnet_dim=dims.postnet_dim, postnet_layers=dims.postnet_layers, n_mel=dims.n_mel, max_seq_len=dims.max_seq_len))
    return out


def _oracle_state(dims):
    import torch
    from meta_tts_amd import synth
    params = {k: torch.from_numpy(v.copy()) for k, v in synth.make_params(dims, 0, weight_scale=WEIGHT_SCALE).items()}
    for k, v in params.items():
        if not k.endswith(("position_enc", "pitch_bins", "energy_bins")):
            v.requires_grad_(True)
    buffers = {k: torch.from_numpy(v.copy()) for k, v in synth.make_buffers(dims).items()}
    names = [k for k, v in params.items() if v.requires_grad]
    return params, buffers, names


def _cpu_partition(nproc):
    """Host hardware threads this process may use, grouped by physical core (sysfs topology) and dealt to `nproc` task processes in
    contiguous runs of whole cores of one package.  Each entry: the cpu ids of the group ordered one-hardware-thread-per-core first,
    then the SMT siblings — so that the first t ids are the right affinity set for t <= cores threads."""
    allowed = sorted(os.sched_getaffinity(0))
    cores = {}
    for c in allowed:
        try:
            with open(f"/sys/devices/system/cpu/cpu{c}/topology/thread_siblings_list") as fh:
                sib = fh.read().strip()
            with open(f"/sys/devices/system/cpu/cpu{c}/topology/physical_package_id") as fh:
                pkg = int(fh.read().strip())
            first = int(sib.replace("-", ",").split(",")[0])
        except (OSError, ValueError):
            pkg, first = 0, c
        cores.setdefault((pkg, first), []).append(c)
    phys = [sorted(v) for _, v in sorted(cores.items())]
    per = max(1, len(phys) // nproc)
    groups = []
    for k in range(nproc):
        mine = phys[k * per:(k + 1) * per] or [phys[k % len(phys)]]
        depth = max(len(c) for c in mine)
        groups.append([c[d] for d in range(depth) for c in mine if d < len(c)])
    return groups, len(phys)


def _cpu_task_worker(j, cpus, thread_legs, barrier, out_q, dropout=True, env=None, interop=0):
    """One task process of the concurrent CPU baseline: task j of the meta-batch (5 inner steps + query forward / backward, first
    order, the oracle, dropout as timed), pinned to its own run of physical cores (`cpus`, _cpu_partition) BEFORE torch / OpenMP start
    (OMP_PLACES=cores, OMP_PROC_BIND=close: intra-op thread i sits on core i of the run).  One leg per entry of `thread_legs` (intra-op
    threads); all processes leave the barrier together, the parent clocks the slowest."""
    pinned = False
    for k, v in (env or {}).items():      # allocator / OpenMP runtime settings: in place before torch (and its OpenMP runtime) load
        os.environ[k] = str(v)
    if cpus:
        try:
            os.sched_setaffinity(0, set(cpus))
            os.environ["OMP_PLACES"] = "cores"
            os.environ["OMP_PROC_BIND"] = "close"
            pinned = True
        except OSError:
            pass
    import torch
    torch.set_num_threads(max(thread_legs))
    if interop:
        torch.set_num_interop_threads(int(interop))
    from meta_tts_amd import synth
    from meta_tts_amd.config import ModelDims, default_algorithm_config
    from oracle import fs2_oracle as O
    dims = ModelDims()
    mods = default_algorithm_config()["adapt"]["modules"]
    params, buffers, names = _oracle_state(dims)
    sup, qry = synth.make_task(j)
    tb_s, tb_q = O.to_torch_batch(sup), O.to_torch_batch(qry)
    masks = _oracle_masks(dims, sup, qry, j) if dropout else None
    # warm-up: one inner step's worth (allocator, thread pool)
    lo = O.fs2_loss(tb_s, O.fs2_forward(params, buffers, *tb_s[2:], n_head=(dims.enc_heads, dims.dec_heads), training=True))
    torch.autograd.grad(lo[0], [params[n] for n in names], allow_unused=True)
    times = []
    for thr in thread_legs:
        torch.set_num_threads(thr)
        barrier.wait()
        t0 = time.perf_counter()
        ql, _, _, _ = O.maml_task(params, buffers, tb_s, tb_q, steps=INNER_STEPS, lr=INNER_LR, second_order=False, modules=mods,
                                  n_head=(dims.enc_heads, dims.dec_heads), dropout=masks)
        torch.autograd.grad(ql[0], [params[n] for n in names], allow_unused=True)
        times.append(time.perf_counter() - t0)
        barrier.wait()
    out_q.put((j, times, pinned))


def cpu_baseline_concurrent(thread_legs, pin=True, timeout_s=240.0, dropout=True, env=None, interop=0):
    """BASELINE.md section 3 "all host cores": the 8 tasks of a meta-batch as 8 processes started together, each pinned to 1/8 of the
    box's physical cores (pin=True) and run once per entry of `thread_legs` intra-op threads; one meta-step = the wall time from the
    common start to the LAST process finishing its task (mean + clip + Adam excluded: < 1 %).  Returns one record per leg."""
    import multiprocessing as mp
    host = os.cpu_count() or 8
    groups, n_phys = _cpu_partition(META_BATCH)
    if not pin:
        groups = [None] * META_BATCH
    ctx = mp.get_context("spawn")
    barrier = ctx.Barrier(META_BATCH + 1)
    q = ctx.Queue()
    procs = [ctx.Process(target=_cpu_task_worker, args=(j, groups[j], list(thread_legs), barrier, q, dropout, env, interop), daemon=True) for j in range(META_BATCH)]
    for pr in procs:
        pr.start()
    walls = []
    try:
        for _ in thread_legs:
            barrier.wait(timeout_s)          # every worker has built its task and warmed up
            t0 = time.perf_counter()
            barrier.wait(timeout_s)          # ... and finished it
            walls.append(time.perf_counter() - t0)
        got = [q.get(timeout=timeout_s) for _ in procs]
    finally:
        for pr in procs:
            pr.join(5.0)
            if pr.is_alive():
                pr.terminate()
    per_task = {j: t for j, t, _ in got}
    pinned = all(p for _, _, p in got)
    out = []
    for i, thr in enumerate(thread_legs):
        hw = min(thr, len(groups[0])) if groups[0] else thr
        out.append({"value": 1.0 / walls[i], "unit": "meta-steps/s", "processes": META_BATCH, "threads_per_process": int(thr), "pinned": bool(pin and pinned),
                    "cores": int(min(host, hw * META_BATCH)), "host_cores": int(host), "physical_cores": int(n_phys), "s_per_meta_step": round(walls[i], 3),
                    "slowest_task_s": round(max(t[i] for t in per_task.values()), 3), "fastest_task_s": round(min(t[i] for t in per_task.values()), 3),
                    "sample": f"one whole 8-task meta-step (dropout {'on' if dropout else 'off'}), 8 task processes x {thr} intra-op threads started together (barrier), "
                              + (f"each pinned to its own {len(groups[0])} hardware threads = {max(1, n_phys // META_BATCH)} physical cores (sched_setaffinity, OMP_PLACES=cores, "
                                 f"OMP_PROC_BIND=close), " if (pin and pinned and groups[0]) else "unpinned, ") + "wall time of the slowest = one meta-step"})
    return out


def cpu_baseline(dims, mods, budget_s=25.0, concurrent=True, dropout=True, make_task=None, inner_steps=None, grad_samples=None):
    """Oracle (oracle/fs2_oracle.py) on the host cores.  Sequential leg: whole first-order tasks of the same workload, one after the
    other at the best swept intra-op thread count, until ~budget_s of CPU time is spent; meta-steps/s = 1 / (8 * mean task time).
    Concurrent leg (cpu_baseline_concurrent): the 8 tasks as 8 processes at once — the harder same-box baseline; `value` is the
    FASTER of the two."""
    import torch
    from meta_tts_amd import synth
    from oracle import fs2_oracle as O
    host_cores = os.cpu_count() or torch.get_num_threads()
    make_task = make_task or synth.make_task
    inner_steps = inner_steps or INNER_STEPS
    grad_samples = grad_samples or GRAD_SAMPLES
    params, buffers, names = _oracle_state(dims)
    # intra-op thread sweep on ONE inner step (support forward + backward of task 0): torch's default (= every hardware thread)
    # is not the fastest setting for these GEMM sizes; the whole-task timing below runs at the best count found
    sup0, qry0 = make_task(0)
    tb0 = O.to_torch_batch(sup0)
    sweep = {}
    for nthr in sorted({t for t in (8, 16, 32, 64, 128) if t <= host_cores}):
        if sweep and min(sweep.values()) * 2.5 < list(sweep.values())[-1]:
            break  # past the optimum: more threads only thrash on these GEMM sizes (256 threads measured 175 s per step)
        torch.set_num_threads(nthr)
        best = None
        for rep in range(2):
            t0 = time.perf_counter()
            lo = O.fs2_loss(tb0, O.fs2_forward(params, buffers, *tb0[2:], n_head=(dims.enc_heads, dims.dec_heads), max_seq_len=dims.max_seq_len, training=True))
            torch.autograd.grad(lo[0], [params[n] for n in names], allow_unused=True)
            dt1 = time.perf_counter() - t0
            best = dt1 if best is None else min(best, dt1)
        sweep[nthr] = round(best, 3)
    cores = min(sweep, key=sweep.get)
    torch.set_num_threads(cores)
    times, q_ref, g_ref, kinks, mels_ref = [], [], [], [], []
    t_all = time.perf_counter()
    j = 0
    while j < META_BATCH and (time.perf_counter() - t_all) < budget_s:
        sup, qry = make_task(j)
        klog = []
        masks = _oracle_masks(dims, sup, qry, j, steps=inner_steps) if dropout else None   # (untimed: the engine's counter-based masks, so that the oracle runs the TIMED configuration)
        t0 = time.perf_counter()
        ql, _, _, qpreds = O.maml_task(params, buffers, O.to_torch_batch(sup), O.to_torch_batch(qry), steps=inner_steps, lr=INNER_LR,
                                       second_order=False, modules=mods, n_head=(dims.enc_heads, dims.dec_heads), max_seq_len=dims.max_seq_len, dropout=masks,
                                       kink_log=klog)
        del masks
        gr = torch.autograd.grad(ql[0], [params[n] for n in names], allow_unused=True)
        times.append(time.perf_counter() - t0)
        mels_ref.append((qpreds[0].detach().numpy().copy(), qpreds[1].detach().numpy().copy()))   # the fp64 arbiter reads this party's L1 signs off its own output
        q_ref.append([float(x) for x in ql])
        relu_log = [e for e in klog if e[0] != "l1"]
        l1_log = [e for e in klog if e[0] == "l1"]
        kinks.append({"relu_units_below_1e-6": int(sum(c for _, c in relu_log)), "min_abs_preactivation": float(min(m for m, _ in relu_log)),
                      "l1_elements_within_1e-4_of_target": int(sum(a + b for _, a, b in l1_log))})
        by_name = dict(zip(names, gr))
        g_ref.append({n: (by_name[n].detach().numpy().copy() if by_name[n] is not None else None) for n in grad_samples})
        j += 1
    mean_t = float(np.mean(times))
    seq = {"value": 1.0 / (META_BATCH * mean_t), "unit": "meta-steps/s", "cores": int(cores),
           "sample": f"{len(times)} of {META_BATCH} tasks ({inner_steps} inner steps + query fwd/bwd each, first-order, dropout {'on' if dropout else 'off'}, fp32 torch-CPU oracle) one after the other at the "
                     f"best of the swept intra-op thread counts ({cores}); {mean_t:.2f} s/task, clip+Adam excluded (<1%)"}
    conc, conc_err, sweep_c, full_box = None, None, {}, None
    if concurrent:
        # 8 task processes at once, each pinned to its own eighth of the box's physical cores (sched_setaffinity + OMP_PLACES=cores +
        # OMP_PROC_BIND=close), ONE spawn, one leg per intra-op thread count: one thread per physical core of the share (= the whole box without
        # SMT: the `full_box` figure), half, a quarter.  Runtime settings measured on the 128-core / 256-thread box (profiles/r06_cpu_baseline.md):
        # OMP_WAIT_POLICY=PASSIVE + GOMP_SPINCOUNT=0 is the one that matters (idle OpenMP workers of 128 busy threads otherwise spin against each
        # other: pinned x16 16.8 -> 10.7 s per meta-step); glibc malloc with one arena and no trimming (MALLOC_*) is worth 3-7 %; a single inter-op
        # thread changes nothing; OMP_WAIT_POLICY=ACTIVE 26 s.  Even so the torch-CPU oracle does not scale with threads on this workload — pinned x4
        # 6.3 s, x8 7.5 s, x16 10.9 s, x32 (SMT) 24.5 s, unpinned x2 6.7 s — a chain of small ops bound by memory traffic and fork-join, not by
        # FLOPs: `value` is the FASTEST leg (the strongest CPU competitor), `full_box` the leg that occupies every physical core.
        groups, n_phys = _cpu_partition(META_BATCH)
        share = max(1, n_phys // META_BATCH)             # physical cores per task process
        legs = sorted({share, max(1, share // 2), max(1, share // 4)}, reverse=True)
        try:
            for r in cpu_baseline_concurrent(legs, pin=True, dropout=dropout, env=CPU_RUNTIME_ENV):
                sweep_c[f"pinned x{r['threads_per_process']}"] = r["s_per_meta_step"]
                r["runtime_env"] = dict(CPU_RUNTIME_ENV)
                if r["threads_per_process"] == share:
                    full_box = r
                if conc is None or r["value"] > conc["value"]:
                    conc = r
        except Exception as ex:  # noqa: BLE001
            conc_err = f"{type(ex).__name__}: {ex}"
        if conc is not None:
            conc["s_per_meta_step_by_leg"] = sweep_c
    best_leg = "concurrent" if (conc is not None and conc["value"] > seq["value"]) else "sequential"
    top = conc if best_leg == "concurrent" else seq
    return {"value": top["value"], "unit": "meta-steps/s", "cores": int(top["cores"]), "kind": "port", "sample": top["sample"], "leg": best_leg,
            "query_losses": q_ref, "grad_samples": g_ref, "query_mels": mels_ref, "query_pass_kinks": kinks,
            "host_cores": int(host_cores), "thread_sweep_s_per_inner_step": {str(k): v for k, v in sweep.items()},
            "sequential": seq, "concurrent": conc if conc is not None else {"error": conc_err},
            "full_box": ({"value": full_box["value"], "unit": "meta-steps/s", "cores": full_box["cores"], "physical_cores": full_box["physical_cores"],
                          "s_per_meta_step": full_box["s_per_meta_step"], "sample": full_box["sample"]} if full_box is not None else None),
            "note": "value = the FASTER of the two CPU legs (speedup_vs_cpu_baseline is quoted against it); north-star target >= 10x"}


def inference_leg(dims, mods, device, iters=5):
    """BASELINE config 5: 5-shot speaker adaptation (5 first-order inner steps
    on the 5 support utterances of task 0) followed by free-running synthesis (predicted durations) of the 5 query texts
    with the adapted weights in train mode, as the reference's test loop does (base_adaptor.py:170-189), then the MelGAN
    generator on the synthesised mels (lightning/utils.py:16-30; synthetic generator weights).  The random-init
    duration predictor emits ~0 frames, so its output bias is set to ln(8) (about 7 frames per phoneme, LibriTTS-like)."""
    import torch
    from meta_tts_amd import synth
    from meta_tts_amd.engine import Engine
    from meta_tts_amd.vocoder import MelGAN
    sup, qry = synth.make_task(0)
    params = synth.make_params(dims, 0)
    params["variance_adaptor.duration_predictor.linear_layer.bias"][:] = np.log(8.0)
    params["variance_adaptor.duration_predictor.linear_layer.weight"] *= 0.25
    eng = Engine(dims, adapt_modules=mods, max_tasks=1, max_B=5, max_S=80, max_T=1000, device=device)
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    eng.load_params(params)
    eng.set_batches(0, [sup])
    voc = MelGAN(max_B=5, max_T=1000, device=device)
    voc.set_stream(torch.cuda.current_stream().cuda_stream)
    res = {}
    for name, with_adapt, with_voc in (("adapt5_plus_synthesis_plus_vocoder", True, True), ("synthesis_plus_vocoder", False, True),
                                       ("adapt5_plus_synthesis", True, False), ("synthesis_only", False, False)):
        frames = 0
        for it in range(iters + 1):
            if it == 1:
                torch.cuda.synchronize(); t0 = time.perf_counter(); frames = 0
            if with_adapt:
                eng.adapt(INNER_STEPS, INNER_LR, reset=True, fetch_losses=False)
            eng.set_batches(1, [qry[:6]], spk_from=[sup], average_spk=True)
            eng.synthesize(1, use_fast=True, train=True)
            if with_voc:
                # mel_post goes engine -> vocoder in HBM; only the durations (for the lengths) and the waveform cross PCIe
                d, mel_lens, tcap = eng.durations(1, 0)
                ptr, tcap, stride = eng.mel_device(1, 0, postnet=True)
                wav_dev = torch.empty((len(mel_lens), tcap * voc.hop), device=f"cuda:{device}", dtype=torch.float32)
                voc.mel2wav_device(ptr, stride, len(mel_lens), tcap, np.maximum(mel_lens, 4), wav_dev.data_ptr(), mel_scale=1.0 / np.log(10.0))
                wav = (wav_dev * 32768.0).to(torch.int16).cpu().numpy()          # LightningMelGAN.infer: x max_wav_value -> int16
                wav = [w[: int(l) * voc.hop] for w, l in zip(wav, mel_lens)]
            else:
                d, mel_lens, tcap = eng.durations(1, 0)
            frames += int(mel_lens.sum())
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        last_lens = [int(x) for x in mel_lens]
        res[name] = {"mels_per_sec": round(frames / dt, 1), "ms_per_iter": round(1e3 * dt / iters, 2), "frames_per_iter": frames // iters,
                     "rtf": round(dt / (frames * 256 / 22050.0), 5)}
    eng.close()
    voc.close()
    # roofline of the two inference stages (blk(L) = 2,883,584 L + 512 L^2 MAC per FFT block,
    # F(S, T) = 4 blk(S) + 3 * 393,472 S + 6 blk(T) + 20,480 T + 4,341,760 T MAC per utterance; MelGAN generator 45.15 MMAC per mel frame),
    # algorithmic flops of VALID phonemes / produced frames over the leg's wall time (which also holds the batch upload and the read-backs)
    blk = lambda L: 2883584.0 * L + 512.0 * L * L
    F = lambda S, T: 4 * blk(S) + 3 * 393472.0 * S + 6 * blk(T) + 20480.0 * T + 4341760.0 * T
    syn_flop = 2.0 * sum(F(int(sl), tl) for sl, tl in zip(qry[4], last_lens))
    voc_flop = 2.0 * 45.15e6 * sum(last_lens)
    t_syn = 1e-3 * res["synthesis_only"]["ms_per_iter"]
    t_voc = max(1e-9, 1e-3 * (res["synthesis_plus_vocoder"]["ms_per_iter"] - res["synthesis_only"]["ms_per_iter"]))
    res["roofline"] = {
        "bound": "mfma", "peak": FP32_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s",
        "synthesis_only": {"alg_gflop_per_iter": round(1e-9 * syn_flop, 2), "achieved": round(1e-12 * syn_flop / t_syn, 2),
                           "frac": round(1e-12 * syn_flop / t_syn / FP32_MATRIX_PEAK_TFLOPS, 4)},
        "vocoder": {"alg_gflop_per_iter": round(1e-9 * voc_flop, 2), "achieved": round(1e-12 * voc_flop / t_voc, 2),
                    "frac": round(1e-12 * voc_flop / t_voc / FP32_MATRIX_PEAK_TFLOPS, 4),
                    "time": "synthesis_plus_vocoder - synthesis_only (generator launches + the waveform's int16 conversion and download)"},
        "note": "whole-leg rates (5 utterances per iteration: launch-bound, see DESIGN.md section 8); the vocoder's early layers (<= 64 channels at 128-256x the "
                "mel rate) are HBM-bound, not MFMA-bound"}
    res["note"] = ("5 query utterances per iteration; host->device batch upload, the duration read-back and (vocoder legs) the mel download, "
                   "waveform int16 conversion and download are inside the timed loop (the mel itself stays in HBM); MelGAN generator with synthetic weights "
                   "(~90 MFLOP per mel frame)")
    res["parity"] = {"acoustic_model": "pinned (tests/golden/c5_synth.npz, reference outputs)",
                     "vocoder_legs": "UNPINNED: the MelGAN generator is a torch.hub dependency absent from the reference tree; its published architecture is "
                                     "restated (oracle/melgan_oracle.py) and timed with synthetic weights — a throughput figure, not a parity claim"}
    return res


def frontend_leg(device, iters=10):
    """The two components in front of the acoustic model: the d-vector speaker encoder (speaker_emb: dvec; the reference runs it on
    the CPU before every forward) on 5 utterances x 6 partial utterances of 160 x 40 mels, and the waveform -> log-mel + energy
    front-end on 10 s of 22.05 kHz audio (filter 1024 / hop 256 / 80 mels).  Host buffers in, host buffers out (PCIe inside)."""
    import torch
    from meta_tts_amd.audio import stft as S
    from meta_tts_amd.speaker_encoder import DVectorEncoder
    g = np.random.RandomState(0)
    res = {}
    enc = DVectorEncoder(max_partials=64, max_utts=8, device=device)
    mels = g.standard_normal((30, 160, 40)).astype(np.float32)
    slices = [slice(6 * i, 6 * i + 6) for i in range(5)]
    enc.embed(mels, slices)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters):
        enc.embed(mels, slices)
    dt = (time.perf_counter() - t0) / iters
    flop = 30 * 160 * 2.0 * (4 * 256 * (40 + 256) + 2 * 4 * 256 * (256 + 256))
    res["dvector_5utt_30partials"] = {"ms_per_call": round(1e3 * dt, 3), "partials_per_sec": round(30 / dt, 1), "gflop": round(flop * 1e-9, 2)}
    enc.enable_training()
    dout = g.standard_normal((5, 256)).astype(np.float32)
    enc.embed_train(mels, slices); enc.backward(dout)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters):
        enc.embed_train(mels, slices); enc.backward(dout)
    torch.cuda.synchronize()
    res["dvector_train_fwd_bwd"] = {"ms_per_call": round(1e3 * (time.perf_counter() - t0) / iters, 3)}
    enc.close()
    st = S.TacotronSTFT(1024, 256, 1024, 80, 22050, 0, None, max_samples=22050 * 11, device=device)
    wav = np.clip(0.3 * g.standard_normal((1, 22050 * 10)), -1, 1).astype(np.float32)
    st.mel_spectrogram(wav)
    t0 = time.perf_counter()
    for _ in range(iters):
        st.mel_spectrogram(wav)
    dt = (time.perf_counter() - t0) / iters
    T = wav.shape[1] // 256 + 1
    res["mel_frontend_10s_audio"] = {"ms_per_call": round(1e3 * dt, 3), "audio_seconds_per_sec": round(10.0 / dt, 1), "frames": T,
                                     "gflop": round(2.0 * T * (1026 * 1024 + 80 * 513) * 1e-9, 2)}
    st.close()
    return res


def mel_l1_leg(dims, device):
    """The metric's third component: mel L1 of this path against the REFERENCE model's output on config C1 (one LibriTTS-shaped
    utterance, S = 80, T = 555), from the committed fixture tests/golden/c1_forward.npz (produced by importing the reference,
    tests/golden/make_golden.py).  Eval mode and train mode (BatchNorm batch statistics); gate 1e-4 (BASELINE north_star)."""
    from meta_tts_amd import synth
    from meta_tts_amd.engine import Engine
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "golden", "c1_forward.npz")
    if not os.path.exists(path):
        return None
    g = np.load(path)
    eng = Engine(dims, adapt_modules=[], max_tasks=1, max_B=1, max_S=80, max_T=555, device=device)
    eng.load_params(synth.make_params(dims, 0))
    eng.set_batches(0, [synth.make_batch(0, 1)])
    out = {"gate": 1e-4, "config": "C1 (1 utterance, S=80, T=555, fp32)", "source": "tests/golden/c1_forward.npz (reference model output)"}
    for train, key, name in ((False, "mel_post", "eval"), (True, "train_mel_post", "train_mode_batchnorm")):
        eng.forward(0, train=train)
        out[name] = float(np.abs(eng.outputs(0, 0)["mel_post"] - g[key]).mean())
    eng.close()
    return out


BF16_MATRIX_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA peak (MI355X_MICROARCH.md; AMD's headline figure includes 2:1 sparsity)


def baseline_c2_leg(dims, device, noam_lr, trn, iters=8, modes=("fp32", "bf16")):
    """BASELINE config 2: multi-task baseline (algorithm=baseline: no inner loop, baseline.py:25-36) on ONE synthetic
    LibriTTS-shaped batch of 16 utterances — forward + backward + clip + Adam per step, dropout on.  Timed in both numerics modes:
    "bf16" is the configuration as BASELINE.json states it (bf16 operands on v_mfma_f32_32x32x16_bf16, fp32 accumulation, fp32
    master weights / optimizer / normalisations; mtts_set_numerics), "fp32" the reference's own arithmetic (main.py:110-112 passes no
    `precision=`) and the parity mode.  Each mode's GEMM launches are timed once more with HIP events for its roofline."""
    import torch
    from meta_tts_amd import synth
    from meta_tts_amd.engine import Engine
    batch = synth.make_batch(0, 16)
    eng = Engine(dims, adapt_modules=(), max_tasks=1, max_B=16, max_S=80, max_T=int(batch[8]), device=device)
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    eng.set_batches(0, [batch])
    frames = int(np.asarray(batch[7]).sum())
    res = {"workload": "C2: algorithm=baseline, batch 16 (sum T = %d frames), fwd + bwd + clip + Adam" % frames}
    eng.set_dropout(True, 99)
    for mode in modes:
        peak = FP32_MATRIX_PEAK_TFLOPS if mode == "fp32" else BF16_MATRIX_PEAK_TFLOPS
        eng.load_params(synth.make_params(dims, 0))
        eng.reset_optimizer()
        eng.set_numerics(mode)
        for it in range(iters + 2):
            if it == 2:
                torch.cuda.synchronize(); t0 = time.perf_counter()
            eng.plain_grad(0, 1.0, fetch_losses=False)
            eng.outer_update(lr=noam_lr(it, dims.d_model, trn["warm_up_step"], trn["anneal_steps"], trn["anneal_rate"]), betas=tuple(trn["betas"]),
                             eps=trn["eps"], weight_decay=trn["weight_decay"], max_norm=trn["grad_clip_thresh"])
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / iters
        rows = gemm_profile(eng, lambda: eng.plain_grad(0, 1.0, fetch_losses=False))
        tot_ms, tot_fl = sum(r[2] for r in rows), sum(r[3] for r in rows)
        dom = max(rows, key=lambda r: r[2])
        res[mode] = {"steps_per_sec": round(1.0 / dt, 3), "ms_per_step": round(1e3 * dt, 2), "frames_per_sec": round(frames / dt, 1),
                     "alg_tflop_per_step": round(tot_fl / 1e12, 3), "whole_step_tflops": round(tot_fl / dt / 1e12, 2),
                     "roofline": {"bound": "mfma", "peak": peak, "unit": "TFLOP/s", "kernel": dom[0], "launches": int(dom[1]),
                                  "achieved": round(dom[3] / (dom[2] * 1e-3) / 1e12, 2) if dom[2] > 0 else 0.0,
                                  "frac": round(dom[3] / (dom[2] * 1e-3) / 1e12 / peak, 4) if dom[2] > 0 else 0.0,
                                  "all_gemm_ms": round(tot_ms, 3), "all_gemm_achieved": round(tot_fl / (tot_ms * 1e-3) / 1e12, 2) if tot_ms > 0 else 0.0,
                                  "all_gemm_frac": round(tot_fl / (tot_ms * 1e-3) / 1e12 / peak, 4) if tot_ms > 0 else 0.0,
                                  "all_gemm_ms_note": "sum over ALL streams (weight-gradient and predictor GEMMs run on side streams beside the critical one), so it may exceed ms_per_step",
                                  "per_kernel": {r[0]: {"launches": int(r[1]), "ms": round(r[2], 3), "tflop": round(r[3] / 1e12, 4)} for r in rows if r[1] > 0}}}
    eng.set_numerics("fp32")
    if "fp32" in res and "bf16" in res:
        res["bf16"]["speedup_vs_fp32"] = round(res["fp32"]["ms_per_step"] / res["bf16"]["ms_per_step"], 2)
    if "bf16" in res:
        res["bf16"]["parity"] = "tests/test_bf16_mode.py: kernel vs bf16-rounded operands (fp32-roundoff bound); C2 losses / sampled gradients vs the fp32 oracle at the stated bf16 tolerances"
    eng.close()
    return res


def gemm_profile(eng, run, sites=False):
    """Run `run()` once with every GEMM launch of this handle timed by HIP events on its stream; rows of
    (kernel, launches, ms, algorithmic flops, algorithmic bytes).  sites: also the per-launch records (kernel name, longest K-loop,
    microseconds, algorithmic GFLOP) — the library's MTTS_GEMM_DUMP csv, read back."""
    import tempfile
    import torch
    dump, user_dump = None, os.environ.get("MTTS_GEMM_DUMP")
    if sites:
        fd, dump = tempfile.mkstemp(suffix=".csv", prefix="mtts_gemm_")
        os.close(fd)
        os.unlink(dump)
        os.environ["MTTS_GEMM_DUMP"] = dump
    eng.profile_gemm(True)
    run()
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    else:
        eng.synchronize()          # (--selftest-emu: the SIMT emulator on CPU)
    rep = eng.profile_report()   # {kernel name as rocprofv3 prints it: [launches, ms, flops, bytes]}
    eng.profile_gemm(False)
    rows = [(name, r[0], r[1], r[2], r[3]) for name, r in rep.items()]
    if not sites:
        return rows
    os.environ.pop("MTTS_GEMM_DUMP", None)
    if user_dump is not None:   # (a caller's own dump request: restored for the legs that follow; this leg's records go to the temp file)
        os.environ["MTTS_GEMM_DUMP"] = user_dump
    recs = []
    try:
        import csv
        with open(dump) as f:
            for r in csv.DictReader(f):
                recs.append((eng.lib.mtts_profile_kernel_name(int(r["kind"])).decode(), int(r["K"]), float(r["us"]), float(r["gflop"]), int(r.get("site", 0)),
                             int(r.get("ctx", 0))))
        if os.environ.get("MTTS_BENCH_KEEP_SITES"):   # (tools/gemm_sites.py reads the per-launch records of the roofline leg from here)
            import shutil
            shutil.copy(dump, os.environ["MTTS_BENCH_KEEP_SITES"])
        os.unlink(dump)
    except Exception:  # noqa: BLE001
        recs = []
    return rows, recs


SITE_CLASSES = {0: "other (mel_linear, embeddings)", 1: "encoder FFT blocks", 2: "decoder FFT blocks", 3: "PostNet (k=5 convolutions)", 4: "variance predictors"}


def launch_classes(recs, kernel, dims):
    """The dominant kernel's launches by call-site class (the library tags every launch record with the model part that issued it; inside
    the FFT blocks the k=9 convolution launches are told apart by their K-loop): launches, ms, achieved TFLOP/s and fraction of the fp32
    matrix peak per class — a PostNet pair (flops of valid frames, rows of the padded rectangle) and a k=9 FFT-block convolution do not
    hide behind one average."""
    k9 = (dims.k1 * dims.d_model, dims.k1 * dims.d_ff)
    agg = {}
    for name, K, us, gf, site, _ctx in recs:
        if name != kernel:
            continue
        label = SITE_CLASSES.get(site, "site %d" % site)
        if site in (1, 2):
            label += ": k=%d convolution launches (K = %d / %d)" % (dims.k1, k9[0], k9[1]) if K in k9 else ": other launches"
        a = agg.setdefault(label, [0, 0.0, 0.0])
        a[0] += 1; a[1] += us; a[2] += gf
    out = {}
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        tf = a[2] / a[1] * 1e3 if a[1] > 0 else 0.0   # GFLOP / us = 1e15 flop/s = 1000 TFLOP/s
        out[k] = {"launches": a[0], "ms": round(a[1] * 1e-3, 2), "achieved": round(tf, 2), "frac": round(tf / FP32_MATRIX_PEAK_TFLOPS, 4)}
    return out


def stream_split(recs):
    """The GEMM launches of the meta-step by the stream that carried them.  A launch is timed by HIP events on its OWN stream: what runs on a side
    stream (the encoder run-ahead of the inner steps, deferred weight gradients) runs UNDER the main stream's launches, so its event time is
    stretched by the sharing and overlaps theirs — `all_gemm.ms_per_meta_step` is the plain sum over all three streams (it can exceed the
    step), the main stream's share is the part of it that sits on the step's critical path."""
    names = {0: "main stream", 1: "weight-gradient side stream", 2: "run-ahead side stream"}
    agg = {}
    for _name, _K, us, gf, _site, ctx in recs:
        a = agg.setdefault(names.get(ctx, "stream %d" % ctx), [0, 0.0, 0.0])
        a[0] += 1; a[1] += us; a[2] += gf
    out = {}
    for k, a in agg.items():
        tf = a[2] / a[1] * 1e3 if a[1] > 0 else 0.0
        out[k] = {"launches": a[0], "ms": round(a[1] * 1e-3, 2), "alg_tflop": round(a[2] * 1e-3, 3), "achieved": round(tf, 2),
                  "frac": round(tf / FP32_MATRIX_PEAK_TFLOPS, 4)}
    return out


def roofline_of(rows, pmc_key=None):
    dom = max(rows, key=lambda r: r[2])
    tot_ms = sum(r[2] for r in rows)
    tot_fl = sum(r[3] for r in rows)
    ach = dom[3] / (dom[2] * 1e-3) / 1e12 if dom[2] > 0 else 0.0
    # HBM bytes per launch of the dominant kernel: PMC counters cannot be read in-process, so this is the figure of the
    # committed separate `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes over this same command (profiles/).
    traffic, traffic_src, mfma_busy = None, None, None
    here = os.path.dirname(os.path.abspath(__file__))
    for name in ("r06_pmc_hbm.json", "r05_pmc_hbm.json", "r04_pmc_hbm.json", "r03_pmc_hbm.json"):
        pmc_path = os.path.join(here, "profiles", name)
        if pmc_key is not None and os.path.exists(pmc_path):
            with open(pmc_path) as f:
                table = json.load(f).get(pmc_key, {})
            # the profiler's name may carry trailing template arguments the launcher's kind name leaves out ("..., 4>" vs "..., 4, false>"): the
            # exact symbol first, then the one symbol it prefixes, and only then the family aggregate (which averages over OTHER instantiations
            # too — rounds 5's line quoted that aggregate for the dominant kernel: 288 MB instead of its own 437 MB per launch)
            k = table.get(dom[0])
            if not k and dom[0].endswith(">"):
                pre = [v for kk, v in table.items() if kk.startswith(dom[0][:-1] + ",")]
                k = pre[0] if len(pre) == 1 else None
            k = k or table.get(dom[0].split("<")[0])
            if k:
                traffic, traffic_src, mfma_busy = k.get("hbm_bytes_per_launch"), "profiles/" + name, k.get("mfma_busy_frac")
                break
    launches = max(dom[1], 1)
    return {"bound": "mfma", "achieved": round(ach, 2), "peak": FP32_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(ach / FP32_MATRIX_PEAK_TFLOPS, 4), "traffic": traffic, "traffic_unit": "bytes/launch", "traffic_source": traffic_src,
            "alg_bytes_per_launch": round(dom[4] / launches), "traffic_over_alg_bytes": round(traffic / (dom[4] / launches), 2) if (traffic and dom[4] > 0) else None,
            "mfma_pipe_busy_frac_pmc": mfma_busy, "kernel": dom[0],
            "launches": int(dom[1]), "avg_launch_us": round(1e3 * dom[2] / launches, 2),
            "alg_gflop_per_launch": round(dom[3] / launches / 1e9, 3),
            "all_gemm": {"ms_per_meta_step": round(tot_ms, 2), "alg_tflop_per_meta_step": round(tot_fl / 1e12, 3),
                         "achieved": round(tot_fl / (tot_ms * 1e-3) / 1e12, 2) if tot_ms > 0 else 0.0,
                         "frac": round(tot_fl / (tot_ms * 1e-3) / 1e12 / FP32_MATRIX_PEAK_TFLOPS, 4) if tot_ms > 0 else 0.0,
                         "per_kernel": {r[0]: {"launches": int(r[1]), "ms": round(r[2], 2), "tflop": round(r[3] / 1e12, 3),
                                               "alg_gbytes": round(r[4] / 1e9, 3)} for r in rows}}}


def replica_digest(eng, dims):
    """CRC32 of everything a replica carries from one meta-step to the next: parameters, Adam m / v, PostNet BatchNorm running buffers."""
    import zlib
    crc = {"theta": 0, "adam_m": 0, "adam_v": 0, "bn_buffers": 0}
    for name in eng.params:
        for key, which in (("theta", 0), ("adam_m", 4), ("adam_v", 5)):
            crc[key] = zlib.crc32(np.ascontiguousarray(eng.export(name, which)).tobytes(), crc[key])
    for i in range(dims.postnet_layers):
        for a in eng.get_bn_buffers(i)[:2]:
            crc["bn_buffers"] = zlib.crc32(np.ascontiguousarray(a).tobytes(), crc["bn_buffers"])
    return crc


def _tiny_dims():
    """--selftest-emu: a small architecture (same code paths, tiny GEMMs) — the model of the CPU test-suite's emulator tests."""
    from meta_tts_amd.config import ModelDims, default_model_config, default_preprocess_config
    mc = default_model_config()
    mc["transformer"].update(dict(encoder_layer=1, decoder_layer=2, encoder_hidden=32, decoder_hidden=32, conv_filter_size=64, encoder_head=2, decoder_head=2))
    mc["variance_predictor"].update(dict(filter_size=32))
    mc["variance_embedding"]["n_bins"] = 16
    mc["max_seq_len"] = 64
    mc["_postnet_dim"] = 48
    pc = default_preprocess_config()
    pc["preprocessing"]["mel"]["n_mel_channels"] = 32
    return ModelDims(mc, pc, n_speaker=12, vocab=40)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-cpu-concurrent", action="store_true", help="skip the 8-process concurrent leg of the CPU baseline")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-inference", action="store_true")
    ap.add_argument("--no-frontend", action="store_true", help="skip the d-vector encoder / mel front-end timing")
    ap.add_argument("--no-baseline-c2", action="store_true", help="skip the extra C2 (multi-task baseline, batch 16) measurement")
    ap.add_argument("--order", type=int, default=1, choices=(1, 2),
                    help="MAML order of the timed meta-step: 1 = BASELINE config C3 (first-order), 2 = the reference's training mode / config C4")
    ap.add_argument("--no-second-order", action="store_true", help="skip the extra second-order measurement")
    ap.add_argument("--no-ar-overlap", action="store_true", help="one blocking all-reduce after the backward instead of the bucketed, overlapped exchange")
    ap.add_argument("--no-dropout", action="store_true", help="parity configuration (dropout = identity) instead of train-mode dropout")
    ap.add_argument("--resident-batches", action="store_true",
                    help="upload the batches once before the timed region instead of every step (round-1 behaviour; the default re-ingests the "
                         "host batches inside every timed step, as training does)")
    ap.add_argument("--emulate-world", type=int, default=0,
                    help="diagnostic: run ONE process with the task share of rank 0 of this many ranks (no collective) to see the per-rank "
                         "step time of an N-GPU run on a 1-GPU box; the JSON line is marked emulated and is not a bench result")
    ap.add_argument("--selftest-spawn", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--selftest-emu", action="store_true", help=argparse.SUPPRESS)   # tests/test_bench_spawn.py: this file's N-rank flow on CPU (gloo + SIMT emulator, tiny model)
    args = ap.parse_args()

    rc = maybe_self_launch(args)
    if rc is not None:
        raise SystemExit(rc)
    if args.selftest_spawn:
        return spawn_selftest(args)

    import torch
    import torch.distributed as dist
    import __graft_entry__ as ge
    from meta_tts_amd import synth
    from meta_tts_amd.config import ModelDims, default_algorithm_config, default_train_config
    from meta_tts_amd.engine import Engine

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:  # never print an n_gpus that is not what was asked for
        raise SystemExit(f"bench.py: {world} rank(s) running but --gpus {args.gpus} requested")
    n = world
    assert META_BATCH % n == 0, "the 8-task meta-batch must split evenly over the ranks"
    emu = bool(args.selftest_emu)       # CPU self-test of THIS flow (not a bench result): gloo ranks, the kernels behind the SIMT emulator, a tiny model
    import datetime
    if emu:
        lib_path, dev = ge.build_emulator(), "cpu"
        if n > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("gloo", rank=rank, world_size=n)
    else:
        lib_path, dev = None, f"cuda:{local_rank}"
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
        if torch.cuda.device_count() < n:
            raise SystemExit(f"bench.py: --gpus {n} but only {torch.cuda.device_count()} GPU(s) visible")
        torch.cuda.set_device(local_rank)
        if os.environ.get("BENCH_STREAM_PRIO"):   # A/B runs: the step's critical stream as a torch stream of this priority (-1 = high)
            torch.cuda.set_stream(torch.cuda.Stream(priority=int(os.environ["BENCH_STREAM_PRIO"])))
        if n > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            # (rank 0 runs its roofline / CPU-baseline / parity legs while the other ranks wait in a barrier: minutes, not the default 10 of the watchdog)
            dist.init_process_group("nccl", rank=rank, world_size=n, timeout=datetime.timedelta(minutes=40))
            assert dist.get_world_size() == n
        if rank == 0:
            ge.build_device()
        if n > 1:
            dist.barrier()

    def dev_sync():
        if emu:
            eng.synchronize()
        else:
            torch.cuda.synchronize()

    alg = default_algorithm_config()
    trn = default_train_config()["optimizer"]
    mods = alg["adapt"]["modules"]
    part = args.emulate_world if (args.emulate_world and n == 1) else n
    local = list(range(rank * META_BATCH // part, (rank + 1) * META_BATCH // part))
    if emu:
        dims = _tiny_dims()
        small = dict(n_mel=dims.n_mel, vocab=dims.vocab, s_range=(5, 13), d_range=(1, 6), first_len=12)
        make_task = lambda j: (synth.make_batch(100 + 2 * j, 3, speaker=2 + j, **small), synth.make_batch(101 + 2 * j, 2, speaker=2 + j, **small))  # noqa: E731
        max_B, max_S, inner_steps = 3, 16, 2
        grad_samples = ("mel_linear.weight", "decoder.layer_stack.1.pos_ffn.w_2.weight", "encoder.layer_stack.0.pos_ffn.w_1.weight",
                        "variance_adaptor.pitch_embedding.weight", "speaker_emb.model.weight")
    else:
        dims = ModelDims()
        make_task, max_B, max_S, inner_steps, grad_samples = synth.make_task, 5, 80, INNER_STEPS, GRAD_SAMPLES
    tasks = [make_task(j) for j in local]
    max_T = max(max(s[8], q[8]) for s, q in tasks)
    eng = Engine(dims, adapt_modules=mods, max_tasks=len(local), max_B=max_B, max_S=max_S, max_T=max_T, device=local_rank if not emu else 0, lib_path=lib_path)
    if not emu:
        eng.set_stream(torch.cuda.current_stream().cuda_stream)
    eng.load_params(synth.make_params(dims, 0, weight_scale=WEIGHT_SCALE))
    eng.set_dropout(not args.no_dropout, DROPOUT_SEED + rank)  # train-mode dropout as in the reference's inner/outer loop (per-rank stream)
    sup_b, qry_b = [t[0] for t in tasks], [t[1] for t in tasks]

    def ingest():
        eng.set_batches(0, sup_b)
        eng.set_batches(1, qry_b, spk_from=sup_b, average_spk=True)

    ingest()
    # the exchange step: RCCL all-reduce of the flat outer gradient, issued by the library on its own stream (mtts_allreduce_outer);
    # torch.distributed only carries the 128-byte unique id.  If the in-library communicator cannot be set up, fall back to
    # torch.distributed's all_reduce on a zero-copy view of the same buffer and say so in the line.
    outer, ar_impl = None, None
    if n > 1:
        # bring-up in lock step so that no rank can be left waiting inside a collective the others never enter: (1) every rank probes the
        # library (ncclGetUniqueId is local), (2) the probes are MIN-reduced, (3) only if all succeeded is the id of rank 0 broadcast
        # and ncclCommInitRank entered, (4) its outcome is MIN-reduced again
        why = ""
        try:
            if emu:
                raise RuntimeError("emulator build: the library's communicator is a loop-back, gloo carries the exchange")
            uid = eng.comm_unique_id()
        except Exception as ex:  # noqa: BLE001
            uid, why = None, str(ex)
        flag = torch.tensor([1 if uid is not None else 0], device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        ok = bool(flag.item())
        if ok:
            ids = [uid if rank == 0 else None]
            dist.broadcast_object_list(ids, src=0)
            try:
                eng.comm_init(ids[0], rank, n)
            except Exception as ex:  # noqa: BLE001
                ok, why = False, str(ex)
            flag = torch.tensor([1 if ok else 0], device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            ok = bool(flag.item())
        if ok:
            ar_impl = "libmtts: ncclAllReduce via dlopen(librccl.so) on the engine stream"
        elif emu:
            import ctypes
            buf = (ctypes.c_float * eng.sync_floats).from_address(eng.outer_grad_ptr())      # emulator: the "device" buffer is host memory
            outer = torch.from_numpy(np.ctypeslib.as_array(buf))
            ar_impl = "torch.distributed (gloo) all_reduce on the emulator's host buffer — CPU self-test"
        else:
            outer = torch.as_tensor(eng.outer_grad_view(), device=f"cuda:{local_rank}")
            ar_impl = f"torch.distributed all_reduce (library communicator unavailable on some rank: {why or 'see other ranks'})"

    step_no = [0]
    ar_events = []
    ar_overlapped = [False]

    class _Clock:
        """Elapsed ms between two points of the stream the exchange is enqueued on (HIP events; wall clock on the emulator)."""
        def __init__(self):
            self.a = self.b = None
        def start(self):
            if emu:
                self.a = time.perf_counter()
            else:
                self.a = torch.cuda.Event(enable_timing=True); self.a.record()
        def stop(self):
            if emu:
                self.b = time.perf_counter()
            else:
                self.b = torch.cuda.Event(enable_timing=True); self.b.record()
        def ms(self):
            return 1e3 * (self.b - self.a) if emu else self.a.elapsed_time(self.b)

    def grad_and_exchange(order=None, timed_ar=False, overlap=True):
        """The gradient call of this rank's tasks + the exchange step (nothing for one rank)."""
        if n > 1 and outer is None and overlap and not args.no_ar_overlap:
            ar_overlapped[0] = eng.arm_allreduce_overlap()   # the buckets leave on the comm stream as the backward completes them
        eng.meta_grad(inner_steps, INNER_LR, 1.0 / META_BATCH, second_order=((order or args.order) == 2), fetch_losses=False)
        if n > 1:
            ck = _Clock() if timed_ar else None
            if ck:
                ck.start()
            if outer is None:
                eng.allreduce_outer()          # overlapped: joins the bucket collectives already in flight (what is timed here is the EXPOSED part);
                                               # otherwise gradient + exchange tail (loss scalars, BatchNorm buffers) in one ncclAllReduce
            else:
                eng.sync_pack(eng.bn_pack_weight(rank, n))
                dist.all_reduce(outer, op=dist.ReduceOp.SUM)
                eng.sync_unpack()
            if ck:
                ck.stop()
                ar_events.append(ck)

    def meta_step(order=None, timed_ar=False):
        if not args.resident_batches:
            ingest()  # host 12-tuples -> HBM + row-space plans, every step (what PL's batch transfer + collate hand-off cost)
        grad_and_exchange(order, timed_ar)
        eng.outer_update(lr=noam_lr(step_no[0], dims.d_model, trn["warm_up_step"], trn["anneal_steps"], trn["anneal_rate"]),
                         betas=tuple(trn["betas"]), eps=trn["eps"], weight_decay=trn["weight_decay"],
                         max_norm=trn["grad_clip_thresh"])
        step_no[0] += 1

    def timed(k, order=None, timed_ar=False):
        dev_sync()
        if n > 1:
            dist.barrier()
        t0 = time.perf_counter()
        for _ in range(k):
            meta_step(order, timed_ar)
        dev_sync()
        if n > 1:
            dist.barrier()
        d = time.perf_counter() - t0
        if n > 1:
            tt = torch.tensor([d], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            d = float(tt.item())
        return d

    for _ in range(args.warmup):
        meta_step()
    dt = timed(args.steps, timed_ar=True)
    # host time to ENQUEUE one meta-step from an idle stream (the device is still running it when the call returns): the host's own cost of the step's
    # ~1 500-1 750 launches.  (Inside the timed loop the host runs ahead until the queue's ring is full, so its loop time there equals the device's.)
    dev_sync()
    t_h = time.perf_counter()
    meta_step()
    host_enqueue_ms = 1e3 * (time.perf_counter() - t_h)
    dev_sync()
    inner_upd = eng.inner_update_launches   # launches of the last timed inner step's SGD update (> 0: module by module behind its backward)
    bucket_agreement = eng.allreduce_bucket_agreement if (n > 1 and outer is None) else None   # 1: the ranks agreed on the bucket table in mtts_comm_init
    ar_ms = None
    ar_launches = eng.allreduce_launches if (n > 1 and ar_overlapped[0]) else (1 if n > 1 else None)
    if ar_events:
        dev_sync()
        ar_ms = float(np.mean([c.ms() for c in ar_events]))  # events on the stream RCCL was enqueued on (torch's current stream)
    # cost of the per-step ingestion alone (host -> HBM + plans), for the record
    dev_sync()
    t_in = time.perf_counter()
    for _ in range(3):
        ingest()
    dev_sync()
    ingest_ms = 1e3 * (time.perf_counter() - t_in) / 3
    so = None
    if args.order == 1 and not args.no_second_order:
        # the same meta-step in the reference's training mode (second-order, config C4 per-GPU work), all ranks
        meta_step(2)
        so_steps = max(1, min(args.steps, 3))
        dso = timed(so_steps, 2)
        so = {"value": round(so_steps / dso, 4), "unit": "meta-steps/s", "ms_per_step": round(1e3 * dso / so_steps, 2), "steps": so_steps,
              "workload": "same 8-task meta-step, second-order MAML (Hessian-vector recursion through the 5 inner steps)"}
    # ---- N > 1: the run verifies itself ---------------------------------------------------------------------------------
    replicas, exchange = None, None
    if n > 1:
        # (1) every rank ends the timed steps with bit-identical weights, Adam moments and BatchNorm buffers: CRC32 of each, all-gathered
        if emu and rank == n - 1 and os.environ.get("MTTS_SELFTEST_BREAK_REPLICA"):   # tests/test_bench_spawn.py: the check must bite
            w = eng.export("mel_linear.bias").copy(); w[0] += 1e-6
            eng.load_params({"mel_linear.bias": w}, strict=False)
        dig = replica_digest(eng, dims)
        allg = [None] * n
        dist.all_gather_object(allg, dig)
        same = all(d == allg[0] for d in allg)
        replicas = {"bit_identical": bool(same), "crc32": {k: "%08x" % v for k, v in allg[0].items()}, "ranks_compared": n,
                    "what": "CRC32 over every parameter tensor, Adam m, Adam v and the PostNet BatchNorm running buffers after the timed steps, one digest per rank"}
        if not same:
            raise SystemExit(f"bench.py: replicas diverged after the timed steps: {allg}")
        # (2) the all-reduced outer gradient equals the one a single handle computes for the whole 8-task meta-batch.  Dropout off for this step only
        # (a rank's masks are keyed by its own seed and the task's position in ITS launch group, a lone handle's by position 0..7), initial weights
        init = synth.make_params(dims, 0, weight_scale=WEIGHT_SCALE)
        eng.load_params(init)
        eng.set_dropout(False, 0)
        ingest()
        grad_and_exchange(1)
        dev_sync()
        reduced = {name: eng.export(name, 1).astype(np.float64) for name in grad_samples}
        synced = np.asarray(eng.synced_losses(), np.float64)
        if rank == 0:
            all_tasks = [make_task(j) for j in range(META_BATCH)]
            e8 = Engine(dims, adapt_modules=mods, max_tasks=META_BATCH, max_B=max_B, max_S=max_S, max_T=max(max(s_[8], q_[8]) for s_, q_ in all_tasks),
                        device=local_rank if not emu else 0, lib_path=lib_path)
            e8.load_params(init)
            e8.set_batches(0, [t[0] for t in all_tasks])
            e8.set_batches(1, [t[1] for t in all_tasks], spk_from=[t[0] for t in all_tasks], average_spk=True)
            q8, _ = e8.meta_grad(inner_steps, INNER_LR, 1.0 / META_BATCH, fetch_losses=True)
            errs = {}
            for name in grad_samples:
                ref8 = e8.export(name, 1).astype(np.float64)
                errs[name] = float(np.abs(reduced[name] - ref8).max() / max(float(np.abs(ref8).max()), 1e-30))
            e8.close()
            loss_err = float(np.abs(synced - np.asarray(q8, np.float64).mean(axis=0)).max() / max(float(np.abs(np.asarray(q8)).max()), 1e-30))
            exchange = {"max_rel": max(errs.values()), "per_tensor": errs, "rtol": EXCHANGE_RTOL, "synced_losses_rel": loss_err,
                        "what": "sampled tensors of the ALL-REDUCED outer gradient of the N-rank run against the outer gradient one handle computes for the whole "
                                "8-task meta-batch (grouped launches) from the same weights, dropout off for this step; max |a - b| / max |b|.  Bound: the grouped-vs-alone "
                                "difference of tests/test_gpu_timed_config.py (other split-K factors / K-groups, amplified by five SGD steps), not bit equality",
                        "ok": bool(max(errs.values()) <= EXCHANGE_RTOL and loss_err <= 1e-4)}
        ok_t = torch.tensor([1 if (exchange is None or exchange["ok"]) else 0], device=dev)
        dist.all_reduce(ok_t, op=dist.ReduceOp.MIN)
        if not bool(ok_t.item()):
            raise SystemExit(f"bench.py: the all-reduced outer gradient disagrees with the single-handle 8-task gradient: {exchange}")
        eng.load_params(init)
        eng.set_dropout(not args.no_dropout, DROPOUT_SEED + rank)
        ingest()
    q_losses = None
    roof = None
    if rank == 0:
        q_losses, _ = eng.meta_grad(inner_steps, INNER_LR, 1.0 / META_BATCH, fetch_losses=True)
    if rank == 0 and not args.no_roofline:
        # (N > 1: rank 0's own launches — its share of the tasks; no collective is armed here, the other ranks wait in the closing barrier)
        rows, recs = gemm_profile(eng, lambda: eng.meta_grad(inner_steps, INNER_LR, 1.0 / META_BATCH, second_order=(args.order == 2), fetch_losses=False), sites=True)
        roof = roofline_of(rows, "kernels" if len(local) == META_BATCH else None)     # (the committed PMC traffic figures are per launch of the 8-task step)
        roof["by_site_class"] = launch_classes(recs, roof["kernel"], dims)
        roof["all_gemm"]["by_stream"] = stream_split(recs)
        roof["tasks_in_the_launches"] = len(local)
        if so is not None:
            rows2 = gemm_profile(eng, lambda: eng.meta_grad(inner_steps, INNER_LR, 1.0 / META_BATCH, second_order=True, fetch_losses=False))
            r2 = roofline_of(rows2, "kernels_second_order" if len(local) == META_BATCH else None)
            so["roofline"] = {k: r2[k] for k in ("bound", "achieved", "peak", "unit", "frac", "kernel", "launches", "avg_launch_us", "alg_gflop_per_launch",
                                                 "alg_bytes_per_launch", "traffic", "traffic_source", "traffic_over_alg_bytes", "mfma_pipe_busy_frac_pmc")}
            so["roofline"]["all_gemm"] = {k: r2["all_gemm"][k] for k in ("ms_per_meta_step", "alg_tflop_per_meta_step", "achieved", "frac")}
            so["whole_step_tflops"] = round(r2["all_gemm"]["alg_tflop_per_meta_step"] * (META_BATCH // len(local)) / (so["ms_per_step"] * 1e-3), 2)
            # SURVEY.md section 8(d) prices a second-order meta-step at 35.5 TFLOP (reverse sweep = 2 x (fwd + bwd) per inner step); the
            # forward-over-reverse HVP executes more (every tangent GEMM is a pair): quote the whole step against the SURVEY figure too
            so["survey_tflop_per_meta_step"] = 35.5
            so["whole_step_tflops_survey_convention"] = round(35.5 / (so["ms_per_step"] * 1e-3), 2)
            so["whole_step_frac_survey_convention"] = round(35.5 / (so["ms_per_step"] * 1e-3) / (FP32_MATRIX_PEAK_TFLOPS * n), 4)
            so["roofline"]["note"] = "achieved = EXECUTED contraction flops of the launches (tangent pairs included) / launch time"
    hbm = None
    if rank == 0 and n == 1 and not emu and not args.no_roofline:
        # the HBM-bound tail of the step, reported as achieved GB/s against the 8 TB/s HBM3E peak: fused clip + Adam
        # (grad-norm reduction + update: 32 B per parameter) timed on its own with events on the launch stream
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 10
        e0.record()
        for _ in range(reps):
            eng.outer_update(lr=1e-9, betas=tuple(trn["betas"]), eps=trn["eps"], weight_decay=trn["weight_decay"], max_norm=trn["grad_clip_thresh"])
        e1.record()
        torch.cuda.synchronize()
        us = 1e3 * e0.elapsed_time(e1) / reps
        nbytes = 32.0 * eng.n_total
        hbm = {"clip_adam": {"us": round(us, 1), "gbytes": round(nbytes / 1e9, 3), "achieved_GBps": round(nbytes / (us * 1e-6) / 1e9, 1), "peak_GBps": 8000.0,
                             "frac": round(nbytes / (us * 1e-6) / 8e12, 3),
                             "bytes_model": "per parameter: 4 (grad, norm pass) + 4 (grad) + 12 (theta, m, v read) + 12 (written)"}}
    cpu = None
    cpu_error = None
    if os.environ.get("MTTS_ABLATE_LN", "0") not in ("", "0") and not args.no_cpu_baseline:
        raise SystemExit("bench.py: MTTS_ABLATE_LN is a timing-only ablation (wrong results): run it with --no-cpu-baseline (no parity gate, no headline line)")
    if rank == 0 and not args.no_cpu_baseline:
        # (N > 1: still rank 0, on the box's host cores, while the other ranks wait in the closing barrier)
        try:
            cpu = cpu_baseline(dims, mods, concurrent=(not args.no_cpu_concurrent) and not emu, dropout=not args.no_dropout, make_task=make_task,
                               inner_steps=inner_steps, grad_samples=grad_samples, budget_s=25.0 if not emu else 120.0)
        except Exception as ex:  # noqa: BLE001
            cpu_error = f"{type(ex).__name__}: {ex}"
    # parity of the TIMED configuration (this rank's grouped tasks, the kernels and launch paths the clock just ran) against the
    # oracle's per-task query losses: the line is refused when they disagree
    parity = None
    if cpu is not None:
        # same handle, same grouped launches, same dropout configuration AND seed as the timed steps (re-seeding restarts the pass
        # counter, so the plan seeds are those of the first timed meta-step; the oracle ran with exactly those masks, _oracle_masks);
        # the weights have moved by the timed Adam steps and the oracle ran on the initial ones, so they are loaded again first
        eng.load_params(synth.make_params(dims, 0, weight_scale=WEIGHT_SCALE))
        eng.set_dropout(not args.no_dropout, DROPOUT_SEED)
        ingest()
        q_parity, _ = eng.meta_grad(inner_steps, INNER_LR, 1.0 / META_BATCH, fetch_losses=True)
        # oracle row j = task j = local[j] on rank 0; an emulated rank holds fewer tasks than the oracle may have run (and the oracle's
        # budget may have stopped short of this rank's share): compare the common prefix
        m = min(len(cpu["query_losses"]), len(local))
        ref = np.asarray(cpu["query_losses"], np.float64)[:m]
        got = np.asarray(q_parity, np.float64)[:m]
        rel = np.abs(got - ref) / np.maximum(np.abs(ref), 1e-6)
        # ... and sampled tensors of the per-task query gradient (what the outer gradient is the mean of).  The loss is piecewise smooth (ReLU in
        # every FFN / predictor, |x| in the two mel terms): where a unit sits inside fp32 noise of its kink two correct fp32 implementations decide
        # differently and their gradients differ by that unit's WHOLE contribution (profiles/r05_dropout_parity.md), so engine-vs-oracle32 alone cannot
        # tell a flip from a bug.  The arbiter (oracle/arbiter.py) evaluates every task in float64 with the same masks and gates
        # err(engine, fp64) <= 3 * err(oracle32, fp64) + 1e-3 per tensor, after taking the L1 signs of the ambiguous elements from each party's OWN
        # mel / mel_post output (exact) and, for a tensor that still fails, switching identified single ReLU units (|pre-activation| < 1e-6 in fp64)
        # whose exactly priced contribution shrinks the residual.  Both errors are reported per task.
        from oracle import arbiter as ARB
        jobs = []
        for jt in range(m):
            eo = eng.outputs(1, jt)
            e_grads = {name: eng.export(name, 2, jt).astype(np.float64) * META_BATCH for name in grad_samples}   # backward ran with grad_scale = 1 / META_BATCH
            o_grads = {name: (g if g is not None else np.zeros_like(e_grads[name])) for name, g in cpu["grad_samples"][jt].items()}
            jobs.append(dict(task=local[jt], group_index=jt, threads=16, dropout_seed=None if args.no_dropout else DROPOUT_SEED, steps=inner_steps, lr=INNER_LR,
                             weight_scale=WEIGHT_SCALE, modules=list(mods), names=list(grad_samples),
                             parties={"engine": {"grads": e_grads, "mel": eo["mel"], "mel_post": eo["mel_post"]},
                                      "oracle32": {"grads": o_grads, "mel": cpu["query_mels"][jt][0], "mel_post": cpu["query_mels"][jt][1]}}))
            if emu:   # the tiny model of the self-test is not what the worker rebuilds from seeds: hand it over
                jobs[-1].update(model=(dims.model_config, dims.preprocess_config, dims.n_speaker, dims.vocab), sup=make_task(local[jt])[0], qry=make_task(local[jt])[1])
        arb_error = None
        try:
            t_arb = time.perf_counter()
            reports = [ARB.synth_task_worker(jb) for jb in jobs] if emu else ARB.run_pool(jobs, processes=min(len(jobs), max(1, (os.cpu_count() or 8) // 16)))
            arb_s = time.perf_counter() - t_arb
            digests = [dict(task=r["task"], **ARB.summarize(r)) for r in reports]
            def worst(key):
                c = [(d[key]["err"], f"task {d['task']}: {d[key]['tensor']}") for d in digests if key in d]
                return max(c) if c else (None, None)
            g_rel, g_worst = worst("engine_gated_max")
            o_rel, o_worst = worst("oracle32_gated_max")
            arb_ok = all(d["pass"] for d in digests)
            parity = {"tasks_checked": int(m), "tasks_grouped_in_the_launches": len(local), "max_rel": float(rel.max()), "rtol": PARITY_RTOL,
                      "what": "per-task query (total, mel, postnet mel, pitch, energy, duration) losses after 5 inner steps, " + ("dropout off" if args.no_dropout else
                              "dropout on (the timed configuration and seed; the oracle applies the engine's counter-based masks, oracle/dropout_masks.py)") + ", vs oracle/fs2_oracle.py",
                      "grad_tensors": list(grad_samples),
                      "grad_gate": reports[0]["gate"], "grad_pass": bool(arb_ok),
                      "grad_err_engine_vs_fp64": g_rel, "grad_err_engine_worst": g_worst, "grad_err_oracle32_vs_fp64": o_rel, "grad_err_oracle32_worst": o_worst,
                      "grad_err_engine_vs_fp64_raw": worst("engine_raw_max")[0], "grad_err_oracle32_vs_fp64_raw": worst("oracle32_raw_max")[0],
                      "grad_err_engine_vs_fp64_l1_signs_only": worst("engine_l1_max")[0], "grad_err_oracle32_vs_fp64_l1_signs_only": worst("oracle32_l1_max")[0],
                      "relu_flips_granted": {"engine": int(sum(d["parties"]["engine"].get("relu_flips_used", 0) for d in digests)),
                                             "oracle32": int(sum(d["parties"]["oracle32"].get("relu_flips_used", 0) for d in digests))},
                      "l1_flips": {"engine": int(sum(d["parties"]["engine"]["l1_flips"] for d in digests)), "oracle32": int(sum(d["parties"]["oracle32"]["l1_flips"] for d in digests))},
                      "grad_what": "sampled per-task query-gradient tensors (first-order outer gradient before the mean) of the engine AND of the fp32 oracle against a float64 "
                                   "evaluation of the same task with the same masks; max |g - g64| / max |g64| per tensor; `raw` = plain; `l1_signs_only` = the L1 signs of "
                                   "elements within 1e-4 of their target taken from the party's own mel / mel_post output; the headline (gated) figure additionally has identified single "
                                   "ReLU units (float64 pre-activation inside 5e-5 of zero) switched for tensors that would otherwise fail, each priced exactly",
                      "arbiter_s": round(arb_s, 1), "per_task": digests}
        except Exception as ex:  # noqa: BLE001
            # the arbiter is infrastructure (8 worker processes, float64 autograd): if IT fails, the gate falls back to the plain comparison with the
            # fp32 oracle at 1e-2 of each tensor's largest entry — no kink allowance of any kind — and the line says so
            arb_error = f"{type(ex).__name__}: {ex}"
            worst_t, worst_e = None, 0.0
            for jb in jobs:
                for name in grad_samples:
                    ge_, go_ = jb["parties"]["engine"]["grads"][name], np.asarray(jb["parties"]["oracle32"]["grads"][name], np.float64)
                    e_ = float(np.abs(ge_ - go_).max() / max(float(np.abs(go_).max()), 1e-30))
                    if e_ > worst_e:
                        worst_e, worst_t = e_, f"task {jb['task']}: {name}"
            arb_ok = worst_e <= 1e-2
            parity = {"tasks_checked": int(m), "tasks_grouped_in_the_launches": len(local), "max_rel": float(rel.max()), "rtol": PARITY_RTOL,
                      "grad_tensors": list(grad_samples), "grad_pass": bool(arb_ok), "arbiter_error": arb_error,
                      "grad_gate": "FALLBACK (the float64 arbiter could not run): max |engine - oracle32| <= 1e-2 * max |oracle32| per sampled tensor, no kink allowance",
                      "grad_max_rel_vs_oracle32": worst_e, "grad_worst": worst_t}
        if not (rel.max() <= PARITY_RTOL) or not arb_ok:
            raise SystemExit(f"bench.py: parity check of the timed configuration failed: {parity}")
    eng.close()
    # auxiliary legs (other BASELINE configs, components beside the hot path): a failure there is reported in the line, it must not
    # cost the headline measurement above
    aux_errors = {}

    def guarded(name, fn, *a):
        try:
            return fn(*a)
        except Exception as ex:  # noqa: BLE001
            aux_errors[name] = f"{type(ex).__name__}: {ex}"
            return None
    infer = None
    if rank == 0 and n == 1 and not emu and not args.no_inference:
        infer = guarded("inference_c5", inference_leg, dims, mods, local_rank)
    mel_l1 = guarded("mel_l1_vs_reference", mel_l1_leg, dims, local_rank) if (rank == 0 and n == 1 and not emu) else None
    c2 = None
    if rank == 0 and n == 1 and not emu and not args.no_baseline_c2:
        c2 = guarded("baseline_c2", baseline_c2_leg, dims, local_rank, noam_lr, trn)
    front = None
    if rank == 0 and n == 1 and not emu and not args.no_frontend and part == n:
        front = guarded("frontend", frontend_leg, local_rank)
    if n > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        ms = 1e3 * dt / args.steps
        line = {"metric": "meta-steps/sec (8-task meta-batch, 5 inner steps)", "value": round(args.steps / dt, 4), "unit": "meta-steps/s",
                "n_gpus": n, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 2), "higher_is_better": True,
                "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                "config": {"workload": ("C3: Meta-TTS MAML first-order" if args.order == 1 else "C4-style: Meta-TTS MAML second-order") + " (algorithm=meta_emb_vad, inner=5, meta-batch=8 tasks x (5 support + 5 query utts)), "
                                       "FastSpeech2 base.yaml, outer mean + clip(1.0) + Adam/Noam", "meta_batch": META_BATCH,
                           "weights": "random init, Linear / Conv1d weight matrices x %g (5 inner steps at lr 1e-3 contractive; see WEIGHT_SCALE)" % WEIGHT_SCALE,
                           "tasks_per_gpu": META_BATCH // part, "inner_steps": inner_steps, "order": "first" if args.order == 1 else "second", "parallelism": f"task-dp{n}",
                           "numerics": "fp32 MFMA (v_mfma_f32_32x32x2_f32)", "dropout": "identity (parity config)" if args.no_dropout else "on (0.2 / 0.5 / 0.5, counter-based masks)",
                           "batch_ingestion": "resident (uploaded once before the timed region)" if args.resident_batches else "inside every timed step (host 12-tuples -> HBM + plans)",
                           "inner_update": ("module by module behind the backward, %d launches per inner step on a stream of its own" % inner_upd)
                                           if inner_upd > 0 else "one launch between the backward and the next forward"},
                **({"emulated_world": part, "note": "diagnostic: rank-0 share of an emulated multi-rank run, no collective"} if part != n else {}),
                "batch_ingest_ms_per_step": round(ingest_ms, 3),
                "host_enqueue_ms_from_idle": round(host_enqueue_ms, 3),
                "rccl_ranks": n if n > 1 else None, "allreduce_impl": ar_impl, "allreduce_ms_per_step": round(ar_ms, 3) if ar_ms is not None else None,
                "allreduce_payload_mbytes": round(4e-6 * eng.n_total, 1) if n > 1 else None,
                "allreduce_overlap": ({"on": bool(ar_overlapped[0]), "collectives_per_step": ar_launches, "bucket_table_agreed_across_ranks": (bucket_agreement == 1) if bucket_agreement is not None else None,
                                       "what": "one ncclAllReduce per module bucket in backward-completion order (PostNet, decoder 5 + mel_linear .. decoder 0, variance adaptor, speaker table, "
                                               "encoder 3 .. 0 + word embedding) + the exchange tail, on a communication stream behind events of the main / weight-gradient streams; "
                                               "allreduce_ms_per_step = what mtts_allreduce_outer still waits for (exposed)"} if n > 1 else None),
                "allreduce_carries": "flat outer gradient + 6 loss scalars (sync_dist mean) + PostNet BatchNorm running buffers (rank 0's, as DDP broadcast_buffers)" if n > 1 else None,
                "replicas": replicas, "outer_gradient_vs_single_handle": exchange,
                "query_total_loss_mean": round(float(q_losses[:, 0].mean()), 5) if q_losses is not None else None}
        if emu:
            line["selftest"] = "emu: CPU self-test of the N-rank flow (gloo + SIMT emulator, tiny model) - NOT a bench result"
        if so is not None:
            line["second_order"] = so
        line["parity_check"] = parity
        if mel_l1 is not None:
            line["mel_l1_vs_reference"] = mel_l1
        if infer is not None:
            line["inference_c5"] = infer
        if c2 is not None:
            line["baseline_c2"] = c2
        if front is not None:
            line["frontend"] = front
        if cpu_error:
            aux_errors["cpu_baseline"] = cpu_error
        if aux_errors:
            line["auxiliary_leg_errors"] = aux_errors
        if roof is not None:
            line["roofline"] = roof
        if hbm is not None:
            line["hbm_bound_kernels"] = hbm
        if cpu is not None:
            cpu = {k: v for k, v in cpu.items() if k not in ("query_losses", "grad_samples", "query_mels")}
            line["cpu_baseline"] = cpu
            line["speedup_vs_cpu_baseline"] = round((args.steps / dt) / cpu["value"], 1)
            if cpu.get("full_box"):
                line["speedup_vs_cpu_baseline_full_box"] = round((args.steps / dt) / cpu["full_box"]["value"], 1)
            if n > 1:
                cpu["note_n_gpus"] = "timed on rank 0's host cores while the other ranks wait in the closing barrier"
        print(json.dumps(line))


if __name__ == "__main__":
    main()
