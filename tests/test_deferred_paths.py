"""The side-stream paths of small plans (engine.h: deferred parameter gradients, encoder run-ahead, predictors on the side stream,
batched predictor GEMMs) and the per-step activation sets of second-order MAML only re-plumb buffers and launch order: with every knob off the engine must produce the SAME outer gradient,
losses and adapted weights, bit for bit.  The knobs are read once per process, so each arm runs in its own interpreter (SIMT emulator
here; `-m gpu` runs the same comparison on the MI355X, where the paths really are concurrent)."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import sys, numpy as np
sys.path.insert(0, {root!r}); sys.path.insert(0, {tests!r})
import __graft_entry__ as ge
from oracle_util import synth, tiny_dims
from meta_tts_amd.engine import Engine
gpu = {gpu}
lib = None if gpu else ge.build_emulator()
dims = tiny_dims()
kw = dict(n_mel=dims.n_mel, vocab=dims.vocab, s_range=(5, 13), d_range=(1, 6), first_len=12)
mods = ["speaker_emb", "variance_adaptor", "decoder", "mel_linear", "postnet"]
tasks = {tasks}
eng = Engine(dims, adapt_modules=mods, max_tasks=tasks, max_B=3, max_S=16, max_T=96, lib_path=lib)
eng.load_params(synth.make_params(dims, 0))
eng.set_dropout(True, 77)
sup = [synth.make_batch(3, 3, speaker=2, **kw), synth.make_batch(4, 2, speaker=5, **kw), synth.make_batch(8, 3, speaker=7, **kw)][:tasks]
qry = [synth.make_batch(5, 2, speaker=2, **kw), synth.make_batch(6, 3, speaker=5, **kw), synth.make_batch(9, 2, speaker=7, **kw)][:tasks]
out = {{}}
for order in (1, 2):
    eng.set_batches(0, sup)
    eng.set_batches(1, qry, spk_from=sup, average_spk=True)
    q, sl = eng.meta_grad(3, 0.02, 0.5, second_order=(order == 2))
    out[f"q{{order}}"] = q
    out[f"s{{order}}"] = sl
    for n in ("mel_linear.weight", "decoder.layer_stack.1.pos_ffn.w_1.weight", "encoder.layer_stack.0.slf_attn.fc.weight",
              "variance_adaptor.pitch_predictor.conv_layer.conv1d_1.conv.bias", "variance_adaptor.energy_predictor.linear_layer.weight",
              "postnet.convolutions.2.0.conv.bias", "decoder.layer_stack.0.slf_attn.layer_norm.weight", "mel_linear.bias",
              "postnet.convolutions.1.0.conv.weight", "postnet.convolutions.4.0.conv.weight"):
        out[f"g{{order}}_" + n] = eng.export(n, 1)
eng.set_batches(0, sup)
eng.adapt(2, 0.02, reset=True, fetch_losses=False)
out["fast"] = eng.export("mel_linear.weight", 3, tasks - 1)
out["upd_launches"] = np.array(eng.inner_update_launches)
np.savez({path!r}, **out)
"""


def _run(tmp_path, tag, env, gpu, tasks=2):
    path = str(tmp_path / f"{tag}.npz")
    code = WORKER.format(root=ROOT, tests=os.path.join(ROOT, "tests"), gpu=gpu, path=path, tasks=tasks)
    e = dict(os.environ)
    e.update(env)
    r = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    return dict(np.load(path))


# (MTTS_SO_KEEP_ACT=0: second-order MAML replays the forward of every inner step in its reverse sweep instead of keeping one activation
# set per step — the same kernels on the same data either way)
OFF = {"MTTS_DEFER_WGRAD": "0", "MTTS_ENC_AHEAD": "0", "MTTS_PRED_SIDE": "0", "MTTS_PRED_EARLY": "0", "MTTS_PRED_BATCH": "0", "MTTS_SINGLE_MULTI": "0",
       "MTTS_SO_KEEP_ACT": "0"}
# kernel-choice knobs: the same mathematics in another summation order (compared at fp32-roundoff tolerance, not bit for bit) — the fused
# attention forward vs grouped GEMM + softmax kernel + grouped GEMM (csrc/attention.h); and the query pass's encoder run-ahead (re-plumbing)
# MTTS_PANEL_ORDER=0: the B-panel-major XCD tile order of under-filled launches off (csrc/gemm.h: xcd_panel_locate) — placement only
# MTTS_PRED_EARLY=0: the phoneme-level predictors' backward at its textual place on the main stream instead of early on the side stream
# MTTS_SO_DEFER_POST=0: second order — the PostNet layers' hv(W) products inside the tangent launches instead of on the side stream (round 5)
# MTTS_LN_FUSE=1: `LayerNorm(dropout(sublayer) + residual)` as the fc / w_2 GEMM's row-complete epilogue (gemm.h: LnFuse, round 5; opt-in: measured
# slower) instead of a launch of its own — the same arithmetic in the same order: bit-identical
# MTTS_ATTN_SORT=0: the attention (sequence, head) groups in batch order instead of longest first (engine.h: build_plan) — independent groups in another
# dispatch order: bit-identical
# MTTS_UPD_OVERLAP=0: the inner SGD step as ONE launch between the backward and the next forward instead of module by module on a stream of
# its own behind the backward (engine.h: upd_ready, round 5) — the same kernel on the same floats: bit-identical
# MTTS_SO_FUSE_DROP=0: second order — the tangent FFT blocks' dropout launches (two in front of the forward's LayerNorm tangents, four behind the backward's) as
# launches of their own instead of riding in the LayerNorm tangent kernels (round 6) — the same masks on the same values: bit-identical
KERNEL_ARMS = [{"MTTS_FUSED_ATTN": "0"}, {"MTTS_ENC_AHEAD_QUERY": "0"}, {"MTTS_PANEL_ORDER": "0"}, {"MTTS_PRED_EARLY": "0"}, {"MTTS_SO_DEFER_POST": "0"},
               {"MTTS_LN_FUSE": "1"}, {"MTTS_UPD_OVERLAP": "0"}, {"MTTS_ATTN_SORT": "0"}, {"MTTS_SO_FUSE_DROP": "0"}, {"MTTS_SO_LN_PART": "0"}, {"MTTS_SO_PRED_SIDE": "0"}, {"MTTS_SO_TABLE_SIDE": "0"}]
# MTTS_SO_TABLE_SIDE=0: second order — the embedding tables' hv (table_grad_kernel) on the critical stream instead of the side stream (from a snapshot where the
# critical stream goes on accumulating into the gradient it reads; round 6): the same kernel on the same floats: bit-identical
# MTTS_SO_PRED_SIDE=0: second order — the hv reductions of the predictors' 256 -> 1 projections on the critical stream instead of the side stream (round 6): the same
# kernels on the same floats: bit-identical
# MTTS_SO_LN_PART=0: second order — hv(gamma) / hv(beta) of every LayerNorm by a two-launch reduction in front of the LayerNorm tangent backward
# (ColArgs mode 5) instead of 8-row partials emitted by that kernel + one fold (round 6): another summation order of the same products — fp32 roundoff


def _compare(tmp_path, gpu):
    # (MTTS_SO_KEEP_GRAD=0 in both arms: with kept primal gradients the reverse sweep reads the dz of the first sweep's LayerNorm backward
    # instead of the tangent kernel's own — equal up to rounding, compared separately below)
    common = {"MTTS_SO_KEEP_GRAD": "0"}
    a = _run(tmp_path, "on", dict(common), gpu)
    b = _run(tmp_path, "off", dict(OFF, **common), gpu)
    c = _run(tmp_path, "keepgrad", {k: v for k, v in common.items() if k != "MTTS_SO_KEEP_GRAD"}, gpu)   # the default configuration
    scale = max(float(np.abs(a[k]).max()) for k in a if k.startswith("g2_"))   # (a conv bias in front of a BatchNorm has a zero gradient: pure rounding noise)
    for k in a:
        if k.startswith("g2_"):   # second-order outer gradient: kept vs recomputed primal gradients
            np.testing.assert_allclose(c[k], a[k], rtol=2e-3, atol=max(2e-5 * float(np.abs(a[k]).max()), 1e-6 * scale), err_msg=k)
        elif not gpu:
            np.testing.assert_array_equal(c[k], a[k], err_msg=k)
    assert set(a) == set(b) and len(a) >= 15
    for k in a:
        assert np.isfinite(a[k]).all(), k
        if gpu and k.startswith(("g", "fast")):
            # on the hardware arm the knobs change the split-K factors / split tails of a few GEMMs (a different, still fixed, summation order)
            np.testing.assert_allclose(a[k], b[k], rtol=2e-4, atol=1e-6 * max(1.0, float(np.abs(b[k]).max())), err_msg=k)
        elif gpu:
            np.testing.assert_allclose(a[k], b[k], rtol=1e-5, err_msg=k)
        else:
            np.testing.assert_array_equal(a[k], b[k], err_msg=k)
    assert np.abs(a["g1_mel_linear.weight"]).max() > 0 and np.abs(a["g2_mel_linear.weight"] - a["g1_mel_linear.weight"]).max() > 0
    for i, arm in enumerate(KERNEL_ARMS):
        d = _run(tmp_path, f"kernel{i}", dict(common, **arm), gpu)
        for k in a:
            if k == "upd_launches":
                # the default ran the inner SGD step module by module (speaker table + variance adaptor, 2 decoder layers, PostNet); the arm, in one pass
                assert int(a[k]) >= 3 and int(d[k]) == (0 if "MTTS_UPD_OVERLAP" in arm else int(a[k])), (arm, a[k], d[k])
            elif "MTTS_LN_FUSE" in arm or "MTTS_UPD_OVERLAP" in arm or "MTTS_ATTN_SORT" in arm or "MTTS_SO_FUSE_DROP" in arm or "MTTS_SO_PRED_SIDE" in arm or "MTTS_SO_TABLE_SIDE" in arm:
                np.testing.assert_array_equal(a[k], d[k], err_msg=f"{arm} {k}")     # (emulator AND hardware: nothing is summed in another order)
            elif ("MTTS_ENC_AHEAD_QUERY" in arm or "MTTS_PANEL_ORDER" in arm or "MTTS_PRED_EARLY" in arm or "MTTS_SO_DEFER_POST" in arm) and not gpu:
                np.testing.assert_array_equal(a[k], d[k], err_msg=f"{arm} {k}")     # (pure re-plumbing: bit-identical)
            else:
                np.testing.assert_allclose(a[k], d[k], rtol=5e-4, atol=2e-6 * max(1.0, float(np.abs(a[k]).max())), err_msg=f"{arm} {k}")


def test_side_stream_paths_change_nothing_emulator(tmp_path):
    _compare(tmp_path, False)


def _compare_beyond_deferred_regime(tmp_path, gpu):
    """Three tasks in the launches — more than the deferred-gradient buffers take: the side-stream predictors (forward and early
    backward) and the encoder run-ahead still apply (MTTS_SIDE_PRED_ALL / MTTS_ENC_AHEAD_ALL) and still change nothing."""
    a = _run(tmp_path, "all_on", {}, gpu, tasks=3)
    b = _run(tmp_path, "all_off", {"MTTS_SIDE_PRED_ALL": "0", "MTTS_ENC_AHEAD_ALL": "0"}, gpu, tasks=3)
    assert set(a) == set(b) and len(a) >= 15
    for k in a:
        assert np.isfinite(a[k]).all(), k
        if gpu:
            np.testing.assert_allclose(a[k], b[k], rtol=2e-4, atol=1e-6 * max(1.0, float(np.abs(b[k]).max())), err_msg=k)
        else:
            np.testing.assert_array_equal(a[k], b[k], err_msg=k)
    # round 6: the FFT blocks' LayerNorm gamma / beta folds on the side stream (per-site partial buffers; MTTS_LN_FOLD_SIDE=0: two colfinal launches
    # per layer inside the critical chain, the shared scratch) and the critical stream's wavefront priority (MTTS_MAIN_PRIO=2: s_setprio 3 in every
    # regime, 0: never) — the same kernels on the same floats in the same order: bit for bit, emulator AND hardware
    c = _run(tmp_path, "fold_main", {"MTTS_LN_FOLD_SIDE": "0", "MTTS_MAIN_PRIO": "0"}, gpu, tasks=3)
    d = _run(tmp_path, "prio_all", {"MTTS_MAIN_PRIO": "2"}, gpu, tasks=3)
    e = _run(tmp_path, "so_drop_launches", {"MTTS_SO_FUSE_DROP": "0"}, gpu, tasks=3)   # (the shared-scratch arm of the tangent blocks' masked copies)
    for k in a:
        np.testing.assert_array_equal(a[k], c[k], err_msg=f"MTTS_LN_FOLD_SIDE=0 {k}")
        np.testing.assert_array_equal(a[k], d[k], err_msg=f"MTTS_MAIN_PRIO=2 {k}")
        np.testing.assert_array_equal(a[k], e[k], err_msg=f"MTTS_SO_FUSE_DROP=0 {k}")


def test_side_stream_paths_beyond_the_deferred_regime_emulator(tmp_path):
    _compare_beyond_deferred_regime(tmp_path, False)



@pytest.mark.gpu
def test_side_stream_paths_change_nothing_gpu(tmp_path):
    _compare(tmp_path, True)


@pytest.mark.gpu
def test_side_stream_paths_beyond_the_deferred_regime_gpu(tmp_path):
    _compare_beyond_deferred_regime(tmp_path, True)
