"""N>1 path on CPU: two gloo ranks, one task each, against one process holding both tasks.  Kernels run
through the SIMT emulator build; what is under test is the sharding contract of bench.py / systems.Trainer:
per-rank 1/total_tasks scaling, SUM all-reduce of the flat outer gradient, identical clip + Adam on every rank."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _host_view(engine):
    """emulator build: the 'device' outer-gradient buffer is host memory -> alias it as a CPU tensor"""
    import ctypes
    buf = (ctypes.c_float * engine.sync_floats).from_address(engine.outer_grad_ptr())   # gradient + exchange tail (losses, BatchNorm buffers)
    return torch.from_numpy(np.ctypeslib.as_array(buf))


def _make(emu_lib, tasks):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_util import tiny_dims
    from meta_tts_amd.config import default_algorithm_config, default_train_config
    from meta_tts_amd.systems import MetaSystem
    dims = tiny_dims()
    pre = dims.preprocess_config
    pre["path"] = {"preprocessed_path": "/nonexistent"}
    return MetaSystem(pre, dims.model_config, default_train_config(), default_algorithm_config(), max_tasks=tasks,
                      max_batch=3, max_src_len=16, max_mel_len=96, lib_path=emu_lib), dims


def _tasks(dims, n):
    from meta_tts_amd import synth
    kw = dict(n_mel=dims.n_mel, vocab=dims.vocab, s_range=(5, 13), d_range=(1, 6), first_len=12)
    return [(synth.make_batch(50 + 2 * j, 3, speaker=2 + j, **kw), synth.make_batch(51 + 2 * j, 2, speaker=2 + j, **kw)) for j in range(n)]


def _worker(rank, world, port, emu_lib, out_dir):
    os.environ["MTTS_EMU_THREADS"] = "2"
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    from meta_tts_amd.systems import Trainer
    sysm, dims = _make(emu_lib, 1)
    tr = Trainer(sysm, outer_grad_tensor=_host_view(sysm.engine))
    tasks = _tasks(dims, world)
    for step in range(2):
        q, s, lr = tr.meta_step([tasks[rank]], total_tasks=world)
    bn = [np.concatenate(sysm.engine.get_bn_buffers(i)[:2]) for i in range(dims.postnet_layers)]
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), w=sysm.engine.export("mel_linear.weight"),
             e=sysm.engine.export("encoder.layer_stack.0.slf_attn.fc.weight"), g=sysm.engine.export("mel_linear.weight", 1), q=q,
             bn=np.concatenate(bn), synced=tr.synced_losses())
    dist.destroy_process_group()


def test_two_rank_meta_step_equals_single_process(tmp_path):
    import __graft_entry__ as ge
    emu_lib = ge.build_emulator()
    port = _free_port()
    mp.start_processes(_worker, args=(2, port, emu_lib, str(tmp_path)), nprocs=2, join=True, start_method="spawn")
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    np.testing.assert_array_equal(r0["w"], r1["w"])  # replicas stay bit-identical
    np.testing.assert_array_equal(r0["g"], r1["g"])
    # ... INCLUDING the BatchNorm running buffers (DDP broadcast_buffers: every rank continues with rank 0's, main.py:32) — each rank saw
    # a different task, so without the exchange tail they would differ
    np.testing.assert_array_equal(r0["bn"], r1["bn"])
    assert np.abs(r0["bn"]).max() > 0
    # ... and the logged losses are the mean over BOTH tasks on every rank (log_dict(sync_dist=True), meta.py:78-79)
    np.testing.assert_array_equal(r0["synced"], r1["synced"])
    np.testing.assert_allclose(r0["synced"], 0.5 * (r0["q"][0] + r1["q"][0]), rtol=1e-6)
    # single process, both tasks grouped in one launch
    from meta_tts_amd.systems import Trainer
    sysm, dims = _make(emu_lib, 2)
    tr = Trainer(sysm)
    tasks = _tasks(dims, 2)
    for step in range(2):
        q, s, lr = tr.meta_step(tasks, total_tasks=2)
    # one process groups both tasks in its launches, a rank runs one: the launcher picks split-K / kernel family by grid
    # fill, so the two computations differ in summation order (fp32 rounding), not in value
    np.testing.assert_allclose(sysm.engine.export("mel_linear.weight", 1), r0["g"], rtol=1e-4, atol=1e-5 * np.abs(r0["g"]).max())
    np.testing.assert_allclose(sysm.engine.export("mel_linear.weight"), r0["w"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(sysm.engine.export("encoder.layer_stack.0.slf_attn.fc.weight"), r0["e"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(q[0], r0["q"][0], rtol=1e-5)
    np.testing.assert_allclose(q[1], r1["q"][0], rtol=1e-5)
