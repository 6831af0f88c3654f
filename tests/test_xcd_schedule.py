"""Task-per-XCD workgroup schedule of 8-task launches (csrc/gemm.h: XcdSched; no reference counterpart — the reference runs its tasks
one after the other, base_adaptor.py:114-131).  It only decides WHICH workgroup computes which tile, so (a) the host-built schedule must
visit every (task, tile) exactly once and load the eight XCDs evenly, (b) an 8-task meta-gradient must be bit-identical with the
schedule on and off.  (b) runs in child interpreters (the switch is read once per process): SIMT emulator here, `-m gpu` on the MI355X."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

import __graft_entry__ as ge
from meta_tts_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_schedule_visits_every_tile_once_and_balances():
    lib = _lib.load(ge.build_emulator())
    g = np.random.RandomState(3)
    cases = [[2100, 1700, 2600, 1900, 2300, 2050, 1800, 2500], [64] * 8, [1, 1, 1, 1, 1, 1, 1, 5000], [300, 0, 310, 290, 305, 0, 280, 330],
             [63, 65, 127, 129, 1, 2, 640, 641]]
    cases += [list(g.randint(1, 3000, 8)) for _ in range(200)]
    for dims8 in cases:
        for groups in (8, 4, 2):   # 4 / 2 groups (the ranks of a 2- / 4-rank job): every group cut into 2 / 4 parts, one per XCD
            dims = dims8[:groups]
            arr = (C.c_int * groups)(*[int(v) for v in dims])
            for cls, tn, upg in ((1, 4, 0), (1, 1, 0), (1, 17, 0), (2, 1, 576), (2, 1, 37), (2, 1, 1)):
                load = C.c_int(0)
                slots = lib.mtts_xcd_schedule_check(arr, groups, cls, tn, upg, C.byref(load))
                assert slots >= 0, (dims, cls, tn, upg)
                units = sum((d + 63) // 64 for d in dims) if cls == 1 else upg * sum(1 for d in dims if d > 0)
                if units >= 256 and min(dims) > 0:   # enough units to balance: the heaviest XCD stays within 15 % of a perfect eighth
                    assert load.value <= 1150, (dims, groups, cls, tn, upg, load.value)
    assert lib.mtts_xcd_schedule_check(None, 8, 1, 1, 0, None) == -1
    assert lib.mtts_xcd_schedule_check((C.c_int * 3)(5, 6, 7), 3, 1, 1, 0, None) == -1   # only 8, 4 or 2 groups


WORKER = r"""
import sys, numpy as np
sys.path.insert(0, {root!r}); sys.path.insert(0, {tests!r})
import __graft_entry__ as ge
from oracle_util import synth, tiny_dims
from meta_tts_amd.engine import Engine
gpu = {gpu}
lib = None if gpu else ge.build_emulator()
dims = tiny_dims()
kw = dict(n_mel=dims.n_mel, vocab=dims.vocab, s_range=(4, 15), d_range=(1, 7), first_len=12)
mods = ["variance_adaptor", "decoder", "mel_linear", "postnet"]
NT = {nt}
eng = Engine(dims, adapt_modules=mods, max_tasks=NT, max_B=3, max_S=16, max_T=112, lib_path=lib)
eng.load_params(synth.make_params(dims, 0))
sup = [synth.make_batch(10 + j, 1 + j % 3, speaker=j, **kw) for j in range(NT)]      # ragged: 1-3 utterances per task
qry = [synth.make_batch(30 + j, 1 + (j + 1) % 3, speaker=j, **kw) for j in range(NT)]
out = {{}}
for order in (1, 2):
    eng.set_batches(0, sup)
    eng.set_batches(1, qry, spk_from=sup, average_spk=True)
    q, sl = eng.meta_grad(2, 0.02, 1.0 / NT, second_order=(order == 2))
    out[f"q{{order}}"] = q
    out[f"s{{order}}"] = sl
    for n in ("mel_linear.weight", "decoder.layer_stack.1.pos_ffn.w_1.weight", "encoder.layer_stack.0.slf_attn.fc.weight",
              "variance_adaptor.pitch_predictor.conv_layer.conv1d_1.conv.weight", "postnet.convolutions.1.0.conv.weight",
              "decoder.layer_stack.0.slf_attn.w_qs.weight", "postnet.convolutions.0.0.conv.bias"):
        out[f"g{{order}}_" + n] = eng.export(n, 1)
np.savez({path!r}, **out)
"""


def _run(tmp_path, tag, env, gpu, nt=8):
    path = str(tmp_path / f"{tag}.npz")
    code = WORKER.format(root=ROOT, tests=os.path.join(ROOT, "tests"), gpu=gpu, path=path, nt=nt)
    e = dict(os.environ)
    e.update(env)
    r = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stderr[-2000:]
    sched_lines = r.stderr.count("xcd_sched cls")
    assert (sched_lines > 100) == (env.get("MTTS_XCD_SCHED") == "1"), sched_lines   # the "on" arm really schedules its launches
    return dict(np.load(path))


def _compare(tmp_path, gpu, nt=8):
    # (MTTS_XCD_SCHED_MIN_GROUPS=2: the opt-in schedules of 4- and 2-task launches too)
    common = {"MTTS_XCD_SCHED_DEBUG": "1", "MTTS_XCD_SCHED_MIN_GROUPS": "2"}
    a = _run(tmp_path, "on", dict(common, MTTS_XCD_SCHED="1"), gpu, nt)
    b = _run(tmp_path, "off", dict(common, MTTS_XCD_SCHED="0"), gpu, nt)
    assert set(a) == set(b) and len(a) >= 16
    for k in a:
        assert np.isfinite(a[k]).all(), k
        np.testing.assert_array_equal(a[k], b[k], err_msg=k)
    assert np.abs(a["g1_mel_linear.weight"]).max() > 0 and np.abs(a["g2_mel_linear.weight"] - a["g1_mel_linear.weight"]).max() > 0


@pytest.mark.parametrize("nt", [8, 4, 2])
def test_schedule_changes_nothing_emulator(tmp_path, nt):
    _compare(tmp_path, False, nt)


@pytest.mark.gpu
@pytest.mark.parametrize("nt", [8, 4])
def test_schedule_changes_nothing_gpu(tmp_path, nt):
    ge.build_device()
    _compare(tmp_path, True, nt)
