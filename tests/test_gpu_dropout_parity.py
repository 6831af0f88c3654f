"""-m gpu: the TIMED configuration with dropout ON against the oracle running the engine's own counter-based masks
(oracle/dropout_masks.py) — VERDICT r04 item 1.  The reference trains with dropout active in the inner and outer loop
(transformer/SubLayers.py:54,90; lightning/model/modules.py:223,235; transformer/Layers.py:133-134); bench.py times that, and
these tests pin it: (a) one full-size plain step, (b) the 8-task grouped first-order meta-gradient, (c) a second-order task
(the Hessian-vector passes replay the inner steps' masks).  Same fp32 tolerances as the dropout-off tests of
tests/test_gpu_model.py / tests/test_gpu_timed_config.py."""
import numpy as np
import pytest
import torch

from oracle_util import O, assert_grad_close, heads, kink_count, synth, torch_buffers, torch_params
from oracle.dropout_masks import DropoutMasks, plan_seed
from meta_tts_amd.config import ModelDims, default_algorithm_config
from meta_tts_amd.engine import Engine

pytestmark = pytest.mark.gpu
DIMS = ModelDims()
MODS = default_algorithm_config()["adapt"]["modules"]
SCALE = 0.5
LR = 0.001
SEED = 20260927
SAMPLED = ["mel_linear.weight", "decoder.layer_stack.5.pos_ffn.w_2.weight", "decoder.layer_stack.0.slf_attn.w_qs.weight",
           "postnet.convolutions.2.0.conv.weight", "variance_adaptor.pitch_predictor.conv_layer.conv1d_1.conv.weight",
           "decoder.layer_stack.3.pos_ffn.w_1.weight", "variance_adaptor.duration_predictor.linear_layer.weight",
           "postnet.convolutions.4.1.weight", "decoder.layer_stack.2.slf_attn.layer_norm.weight"]


@pytest.fixture(scope="module", autouse=True)
def _build():
    import __graft_entry__ as ge
    ge.build_device()
    torch.set_num_threads(16)


@pytest.fixture(scope="module")
def tasks():
    return [synth.make_task(j) for j in range(8)]


def _engine(n_tasks, tasks):
    max_T = max(max(s[8], q[8]) for s, q in tasks)
    eng = Engine(DIMS, adapt_modules=MODS, max_tasks=n_tasks, max_B=5, max_S=80, max_T=max_T)
    eng.load_params(synth.make_params(DIMS, 0, weight_scale=SCALE))
    return eng


def _set(eng, tasks):
    sup, qry = [t[0] for t in tasks], [t[1] for t in tasks]
    eng.set_batches(0, sup)
    eng.set_batches(1, qry, spk_from=sup, average_spk=True)


def test_plain_step_full_size_dropout_on(tasks):
    """(a) C3 task 0's support batch, one plain_grad (forward + loss + backward incl. the encoder) with dropout on: mel / predictions,
    the 6 losses and sampled parameter gradients (encoder tensors included) vs autograd through the oracle with the same masks."""
    sup = tasks[0][0]
    eng = _engine(1, [tasks[0]])
    eng.set_batches(0, [sup])
    eng.set_dropout(True, SEED)
    q = eng.plain_grad(0, 1.0)
    out = eng.outputs(0, 0)
    p = torch_params(DIMS, requires_grad=True, weight_scale=SCALE)
    tb = O.to_torch_batch(sup)
    dm = DropoutMasks(plan_seed(SEED, 1), 0)
    klog = []
    O.KINK_LOG = klog
    try:
        o = O.fs2_forward(p, torch_buffers(DIMS), *tb[2:], n_head=heads(DIMS), training=True, dropout=dm)
    finally:
        O.KINK_LOG = None
    lo = O.fs2_loss(tb, o, kink_log=klog)
    l1 = float(np.abs(out["mel_post"] - o[1].detach().numpy()).mean())
    assert l1 < 1e-4, f"mel L1 vs oracle with dropout on: {l1}"           # the north-star gate, in the timed configuration
    for k, ref in (("mel", o[0]), ("p", o[2]), ("e", o[3]), ("logd", o[4])):
        assert np.abs(out[k] - ref.detach().numpy()).max() < 2e-3, k
    np.testing.assert_allclose(q[0], [float(x) for x in lo], rtol=2e-5)
    names = SAMPLED + ["encoder.layer_stack.0.slf_attn.w_qs.weight", "encoder.layer_stack.3.pos_ffn.w_1.weight", "encoder.src_word_emb.weight"]
    gs = torch.autograd.grad(lo[0], [p[n] for n in names])
    for n, g in zip(names, gs):
        assert_grad_close(eng.export(n, 1), g.numpy(), 1e-3, kink_count(klog), n, atol=1e-7)
    # and the masks matter at these tolerances: the dropout-off oracle is far away
    with torch.no_grad():
        o0 = O.fs2_forward(p, torch_buffers(DIMS), *tb[2:], n_head=heads(DIMS), training=True)
    assert float(np.abs(out["mel_post"] - o0[1].numpy()).mean()) > 1e-2
    eng.close()


def test_eight_grouped_tasks_first_order_dropout_on(tasks):
    """(b) one C3 meta-gradient exactly as bench.py issues it — 8 tasks in every launch, grad_scale 1/8, dropout ON, encoder
    run-ahead drawing the per-step seeds up front — per-task query 6-tuples and support losses for ALL 8 tasks and sampled tensors
    of the outer gradient against the mean of the per-task autograd gradients of O.maml_task with the engine's masks (plan seeds in draw order: 5
    inner steps, then the query pass)."""
    eng = _engine(8, tasks)
    _set(eng, tasks)
    eng.set_dropout(True, SEED)
    q, s = eng.meta_grad(5, LR, 1.0 / 8)
    p = torch_params(DIMS, requires_grad=True, weight_scale=SCALE)
    buf = torch_buffers(DIMS)
    ref_g = {n: np.zeros_like(p[n].detach().numpy()) for n in SAMPLED}
    kinks = 0
    for j, (sup, qry) in enumerate(tasks):
        dms = [DropoutMasks(plan_seed(SEED, k + 1), j) for k in range(6)]
        klog = []
        ql, sl, _, _ = O.maml_task(p, buf, O.to_torch_batch(sup), O.to_torch_batch(qry), steps=5, lr=LR, second_order=False, modules=MODS,
                                   n_head=heads(DIMS), dropout=dms, kink_log=klog)
        kinks += kink_count(klog)
        np.testing.assert_allclose(q[j], [float(x) for x in ql], rtol=2e-3, err_msg=f"query losses of task {j}")
        np.testing.assert_allclose(s[:, j, :], np.array([[float(x) for x in l] for l in sl]), rtol=2e-3, err_msg=f"support losses of task {j}")
        gs = torch.autograd.grad(ql[0], [p[n] for n in SAMPLED])
        for n, g in zip(SAMPLED, gs):
            ref_g[n] += g.numpy() / 8.0
    for n in SAMPLED:                                        # which = 1: the outer gradient (mean over the 8 tasks)
        assert_grad_close(eng.export(n, 1), ref_g[n], 3e-3, kinks, n)
    eng.close()


def test_second_order_task_dropout_on(tasks):
    """(c) second-order MAML (base_adaptor.py:107) with dropout on, tasks 1 and 6 grouped with the other six: the reverse sweep's
    tangent forward / backward must regenerate each inner step's masks (seed_override replay)."""
    eng = _engine(8, tasks)
    _set(eng, tasks)
    eng.set_dropout(True, SEED + 1)
    q, _ = eng.meta_grad(5, LR, 1.0, second_order=True)
    p = torch_params(DIMS, requires_grad=True, weight_scale=SCALE)
    buf = torch_buffers(DIMS)
    names = SAMPLED + ["encoder.layer_stack.0.slf_attn.w_qs.weight"]
    for j in (1, 6):
        sup, qry = tasks[j]
        dms = [DropoutMasks(plan_seed(SEED + 1, k + 1), j) for k in range(6)]
        klog = []
        ql, _, _, _ = O.maml_task(p, buf, O.to_torch_batch(sup), O.to_torch_batch(qry), steps=5, lr=LR, second_order=True, modules=MODS,
                                  n_head=heads(DIMS), dropout=dms, kink_log=klog)
        np.testing.assert_allclose(q[j], [float(x) for x in ql], rtol=2e-3)
        gs = torch.autograd.grad(ql[0], [p[n] for n in names])
        for n, g in zip(names, gs):
            # (second order also differentiates through the inner steps' kinks, which the query-pass log does not see: the L2 fallback applies
            # whenever the query pass itself had one)
            assert_grad_close(eng.export(n, 2, j), g.numpy(), 1.5e-2 if n.startswith("encoder.") else 5e-3, kink_count(klog), f"task {j}: {n}")
    eng.close()
