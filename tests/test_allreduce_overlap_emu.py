"""The overlapped, bucketed exchange (include/mtts.h: mtts_arm_allreduce_overlap; csrc/engine.h: ar_*) on the CPU: the emulator build's
loop-back communicator behaves like `world` ranks holding identical data (SUM = world x local), so a float of the exchange buffer that
is never reduced stays 1 x and one reduced twice becomes world^2 x — the overlapped path must leave EXACTLY what the one-shot
mtts_allreduce_outer leaves (gradient, loss scalars, BatchNorm buffers), bit for bit, for first order, second order, the plain
(baseline) step and the last call of a gradient-accumulation window.  Reference: main.py:30-38 (DDP's bucketed gradient all-reduce
overlapping the backward).  The RCCL leg (world size 1) is tests/test_gpu_c5_training.py."""
import numpy as np
import pytest

import __graft_entry__ as ge
from oracle_util import synth, tiny_dims
from meta_tts_amd.engine import Engine

MODS = ["speaker_emb", "variance_adaptor", "decoder", "mel_linear", "postnet"]
WORLD = 2


@pytest.fixture(scope="module")
def emu_lib():
    return ge.build_emulator()


def _engine(emu_lib, tasks):
    dims = tiny_dims()
    eng = Engine(dims, adapt_modules=MODS, max_tasks=tasks, max_B=3, max_S=16, max_T=96, lib_path=emu_lib)
    eng.load_params(synth.make_params(dims, 0))
    eng.comm_init(eng.comm_unique_id(), 0, WORLD)
    kw = dict(n_mel=dims.n_mel, vocab=dims.vocab, s_range=(5, 13), d_range=(1, 6), first_len=12)
    sup = [synth.make_batch(31 + 2 * j, 3, speaker=2 + j, **kw) for j in range(tasks)]
    qry = [synth.make_batch(32 + 2 * j, 2, speaker=2 + j, **kw) for j in range(tasks)]
    return dims, eng, sup, qry


def _state(eng):
    """everything the exchange moves: the whole outer gradient, the reduced losses, the BatchNorm running buffers"""
    out = {n: eng.export(n, 1).copy() for n in eng.params}
    out["__losses"] = np.array(eng.synced_losses())
    for i in range(5):
        m, v, _ = eng.get_bn_buffers(i)
        out[f"__bn{i}"] = np.concatenate([m, v])
    return out


def _run(eng, sup, qry, kind, overlap, accumulate=False):
    eng.set_dropout(True, 77)                      # same masks in both arms
    for i in range(5):                             # same BatchNorm running buffers going in
        m, v, _ = eng.get_bn_buffers(i)
        eng.set_bn_buffers(i, np.zeros_like(m), np.ones_like(v), 0)
    eng.set_batches(0, sup)
    eng.set_batches(1, qry, spk_from=sup, average_spk=True)
    armed = None
    if accumulate:                                 # first call of a 2-call window: overwrites outer, no exchange
        if kind == "plain":
            eng.plain_grad(1, 0.25, fetch_losses=False)
        else:
            eng.meta_grad(2, 1e-3, 0.25, second_order=(kind == "so"), fetch_losses=False)
        eng.set_grad_accumulation(True)
    if overlap:
        armed = eng.arm_allreduce_overlap()
    if kind == "plain":
        eng.plain_grad(0, 0.5, fetch_losses=False)
    else:
        eng.meta_grad(2, 1e-3, 0.5, second_order=(kind == "so"), fetch_losses=False)
    eng.set_grad_accumulation(False)
    eng.allreduce_outer()
    eng.synchronize()
    return _state(eng), armed, eng.allreduce_launches


@pytest.mark.parametrize("accumulate", [False, True])
@pytest.mark.parametrize("kind", ["fo", "so", "plain"])
def test_overlapped_exchange_equals_one_shot(emu_lib, kind, accumulate):
    dims, eng, sup, qry = _engine(emu_lib, 2)
    ref, _, _ = _run(eng, sup, qry, kind, overlap=False, accumulate=accumulate)
    got, armed, launches = _run(eng, sup, qry, kind, overlap=True, accumulate=accumulate)
    assert armed is True
    assert launches == 3 + dims.dec_layers + dims.enc_layers + 1     # one collective per bucket + the tail
    # the loop-back SUM really doubled things (a world-1 run would make this test vacuous)
    assert float(np.abs(ref["__losses"]).sum()) > 0
    for k in ref:
        np.testing.assert_array_equal(got[k], ref[k], err_msg=k)
    eng.close()


def test_exchange_really_sums_and_unarmed_calls_are_unchanged(emu_lib):
    """world = 2 loop-back vs no exchange at all: every tensor of the outer gradient is exactly doubled by either path; a gradient call
    that was not armed issues no bucket collectives."""
    dims, eng, sup, qry = _engine(emu_lib, 1)
    eng.set_dropout(False)
    eng.set_batches(0, sup)
    eng.set_batches(1, qry, spk_from=sup, average_spk=True)
    eng.meta_grad(2, 1e-3, 1.0, fetch_losses=False)
    local = {n: eng.export(n, 1).copy() for n in eng.params}
    assert eng.arm_allreduce_overlap()
    eng.meta_grad(2, 1e-3, 1.0, fetch_losses=False)
    eng.allreduce_outer()
    for n in eng.params:
        np.testing.assert_array_equal(eng.export(n, 1), 2.0 * local[n], err_msg=n)
    eng.meta_grad(2, 1e-3, 1.0, fetch_losses=False)     # not armed: the one-shot path on the next allreduce_outer
    eng.allreduce_outer()
    for n in eng.params:
        np.testing.assert_array_equal(eng.export(n, 1), 2.0 * local[n], err_msg=n)
    eng.close()


def test_arming_does_not_outlive_a_failed_or_abandoned_gradient_call(emu_lib):
    """the arming is one-shot.  (a) A gradient call that fails BEFORE it reaches the exchange (here: too many inner steps — refused in
    the C ABI) must leave the handle disarmed; (b) mtts_disarm_allreduce_overlap takes an arming back (systems.Trainer calls it in its finally).
    In both cases the next, un-armed gradient call issues NO collectives: its outer gradient is the local one (the loop-back communicator would
    have doubled it)."""
    from meta_tts_amd.engine import MttsError
    dims, eng, sup, qry = _engine(emu_lib, 1)
    assert eng.allreduce_bucket_agreement == 1          # agreed collectively in comm_init
    eng.set_dropout(False)
    eng.set_batches(0, sup)
    eng.set_batches(1, qry, spk_from=sup, average_spk=True)
    eng.meta_grad(2, 1e-3, 1.0, fetch_losses=False)
    local = {n: eng.export(n, 1).copy() for n in eng.params}
    # (a)
    assert eng.arm_allreduce_overlap()
    with pytest.raises(MttsError):
        eng.meta_grad(10 ** 6, 1e-3, 1.0, fetch_losses=False)
    eng.meta_grad(2, 1e-3, 1.0, fetch_losses=False)
    eng.synchronize()
    for n in eng.params:
        np.testing.assert_array_equal(eng.export(n, 1), local[n], err_msg=n)
    # (b)
    assert eng.arm_allreduce_overlap()
    eng.disarm_allreduce_overlap()
    eng.meta_grad(2, 1e-3, 1.0, fetch_losses=False)
    eng.synchronize()
    for n in eng.params:
        np.testing.assert_array_equal(eng.export(n, 1), local[n], err_msg=n)
    eng.close()
