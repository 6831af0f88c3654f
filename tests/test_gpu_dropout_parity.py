"""-m gpu: the TIMED configuration with dropout ON against the oracle running the engine's own counter-based masks
(oracle/dropout_masks.py) — VERDICT r04 item 1.  The reference trains with dropout active in the inner and outer loop
(transformer/SubLayers.py:54,90; lightning/model/modules.py:223,235; transformer/Layers.py:133-134); bench.py times that, and
these tests pin it: (a) one full-size plain step, (b) the 8-task grouped first-order meta-gradient, (c) a second-order task
(the Hessian-vector passes replay the inner steps' masks).  Same fp32 tolerances as the dropout-off tests of
tests/test_gpu_model.py / tests/test_gpu_timed_config.py."""
import numpy as np
import pytest
import torch

from oracle_util import O, check_grads, heads, synth, torch_buffers, torch_params
from oracle import arbiter as ARB
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
    o = O.fs2_forward(p, torch_buffers(DIMS), *tb[2:], n_head=heads(DIMS), training=True, dropout=dm)
    lo = O.fs2_loss(tb, o)
    l1 = float(np.abs(out["mel_post"] - o[1].detach().numpy()).mean())
    assert l1 < 1e-4, f"mel L1 vs oracle with dropout on: {l1}"           # the north-star gate, in the timed configuration
    for k, ref in (("mel", o[0]), ("p", o[2]), ("e", o[3]), ("logd", o[4])):
        assert np.abs(out[k] - ref.detach().numpy()).max() < 2e-3, k
    np.testing.assert_allclose(q[0], [float(x) for x in lo], rtol=2e-5)
    names = SAMPLED + ["encoder.layer_stack.0.slf_attn.w_qs.weight", "encoder.layer_stack.3.pos_ffn.w_1.weight", "encoder.src_word_emb.weight"]
    gs = torch.autograd.grad(lo[0], [p[n] for n in names])
    got = {n: eng.export(n, 1) for n in names}
    ref = {n: g.numpy() for n, g in zip(names, gs)}

    def arbitrate(failing):   # a plain step = a "task" with no inner steps whose query batch is the support batch (mean of 5 identical speaker rows)
        return ARB.arbitrate_task(synth.make_params(DIMS, 0, weight_scale=SCALE), synth.make_buffers(DIMS), sup, sup, modules=MODS, n_head=heads(DIMS),
                                  max_seq_len=DIMS.max_seq_len, steps=0, lr=0.0, masks=[DropoutMasks(plan_seed(SEED, 1), 0)], names=failing,
                                  parties={"engine": {"grads": got, "mel": out["mel"], "mel_post": out["mel_post"]},
                                           "oracle32": {"grads": ref, "mel": o[0].detach().numpy(), "mel_post": o[1].detach().numpy()}})
    check_grads({n: (got[n], ref[n]) for n in names}, 1e-3, arbitrate, atol=1e-7, label="plain step")
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
    mean_g = {n: np.zeros(p[n].shape, np.float64) for n in SAMPLED}
    for j, (sup, qry) in enumerate(tasks):
        dms = [DropoutMasks(plan_seed(SEED, k + 1), j) for k in range(6)]
        ql, sl, _, qp = O.maml_task(p, buf, O.to_torch_batch(sup), O.to_torch_batch(qry), steps=5, lr=LR, second_order=False, modules=MODS,
                                    n_head=heads(DIMS), dropout=dms)
        np.testing.assert_allclose(q[j], [float(x) for x in ql], rtol=2e-3, err_msg=f"query losses of task {j}")
        np.testing.assert_allclose(s[:, j, :], np.array([[float(x) for x in l] for l in sl]), rtol=2e-3, err_msg=f"support losses of task {j}")
        gs = torch.autograd.grad(ql[0], [p[n] for n in SAMPLED])
        ref = {n: g.numpy() for n, g in zip(SAMPLED, gs)}
        got = {n: eng.export(n, 2, j).astype(np.float64) * 8.0 for n in SAMPLED}          # which = 2: task j's own gradient (grad_scale 1/8)
        out = eng.outputs(1, j)

        def arbitrate(failing, j=j, got=got, ref=ref, out=out, qp=qp):
            return ARB.synth_task_worker(dict(task=j, threads=16, dropout_seed=SEED, steps=5, lr=LR, weight_scale=SCALE, modules=MODS, names=failing,
                                              parties={"engine": {"grads": got, "mel": out["mel"], "mel_post": out["mel_post"]},
                                                       "oracle32": {"grads": ref, "mel": qp[0].detach().numpy(), "mel_post": qp[1].detach().numpy()}}))
        check_grads({n: (got[n], ref[n]) for n in SAMPLED}, 3e-3, arbitrate, label=f"task {j}")
        for n in SAMPLED:
            mean_g[n] += got[n] / 8.0
    for n in SAMPLED:                                        # which = 1: the outer gradient IS the mean of the per-task gradients just checked
        np.testing.assert_allclose(eng.export(n, 1), mean_g[n], rtol=0, atol=2e-6 * np.abs(mean_g[n]).max(), err_msg=n)
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
        ql, _, _, qp = O.maml_task(p, buf, O.to_torch_batch(sup), O.to_torch_batch(qry), steps=5, lr=LR, second_order=True, modules=MODS,
                                   n_head=heads(DIMS), dropout=dms)
        np.testing.assert_allclose(q[j], [float(x) for x in ql], rtol=2e-3)
        gs = torch.autograd.grad(ql[0], [p[n] for n in names])
        ref = {n: g.numpy() for n, g in zip(names, gs)}
        got = {n: eng.export(n, 2, j) for n in names}
        out = eng.outputs(1, j)
        T = out["mel_post"].shape[1]       # the query pass's outputs survive the reverse sweep (the arbiter reads the engine's L1 signs off them)
        assert float(np.abs(out["mel_post"] - qp[1].detach().numpy()[:, :T]).mean()) < 1e-4

        def arbitrate(failing, j=j, got=got, ref=ref, out=out, qp=qp):
            # second order in float64 (create_graph through the five inner steps).  A kink of an INNER pass moves the outer gradient only by
            # lr x that unit's contribution (it changes the fast weights, not the query graph), so the query pass's kinks are the ones priced.
            return ARB.synth_task_worker(dict(task=j, threads=16, dropout_seed=SEED + 1, steps=5, lr=LR, weight_scale=SCALE, modules=MODS, names=failing,
                                              second_order=True,
                                              parties={"engine": {"grads": got, "mel": out["mel"], "mel_post": out["mel_post"]},
                                                       "oracle32": {"grads": ref, "mel": qp[0].detach().numpy(), "mel_post": qp[1].detach().numpy()}}))
        tight = {n: (got[n], ref[n]) for n in names if not n.startswith("encoder.")}
        check_grads(tight, 5e-3, arbitrate, label=f"task {j}")
        # (the encoder's q projection only sees the second-order terms — the smallest signal of the set — and five reverse steps amplify
        # summation-order differences most there: tests/test_gpu_timed_config.py)
        check_grads({n: (got[n], ref[n]) for n in names if n.startswith("encoder.")}, 1.5e-2, arbitrate, label=f"task {j}")
    eng.close()


@pytest.mark.slow
def test_eight_tasks_every_sampled_tensor_through_the_fp64_arbiter(tasks):
    """The timed configuration judged the way bench.py's parity_check judges it, for ALL 8 tasks and every sampled tensor whether or not it agrees
    tightly with the fp32 oracle: float64 evaluation per task (8 worker processes), err(engine, fp64) <= 3 * err(oracle32, fp64) + 1e-3."""
    eng = _engine(8, tasks)
    _set(eng, tasks)
    eng.set_dropout(True, SEED)
    eng.meta_grad(5, LR, 1.0 / 8)
    p = torch_params(DIMS, requires_grad=True, weight_scale=SCALE)
    buf = torch_buffers(DIMS)
    jobs = []
    for j, (sup, qry) in enumerate(tasks):
        dms = [DropoutMasks(plan_seed(SEED, k + 1), j) for k in range(6)]
        ql, _, _, qp = O.maml_task(p, buf, O.to_torch_batch(sup), O.to_torch_batch(qry), steps=5, lr=LR, second_order=False, modules=MODS,
                                   n_head=heads(DIMS), dropout=dms)
        gs = torch.autograd.grad(ql[0], [p[n] for n in SAMPLED])
        out = eng.outputs(1, j)
        jobs.append(dict(task=j, threads=16, dropout_seed=SEED, steps=5, lr=LR, weight_scale=SCALE, modules=MODS, names=SAMPLED,
                         parties={"engine": {"grads": {n: eng.export(n, 2, j).astype(np.float64) * 8.0 for n in SAMPLED}, "mel": out["mel"], "mel_post": out["mel_post"]},
                                  "oracle32": {"grads": {n: g.numpy() for n, g in zip(SAMPLED, gs)}, "mel": qp[0].detach().numpy(), "mel_post": qp[1].detach().numpy()}}))
    eng.close()
    reports = ARB.run_pool(jobs, processes=8)
    digests = [ARB.summarize(r) for r in reports]
    assert all(d["pass"] for d in digests), digests
    for d in digests:      # the engine is as close to float64 as the fp32 oracle is (the gate's floor is 1e-3)
        assert d["engine_l1_max"]["err"] <= 3 * d["oracle32_l1_max"]["err"] + 1e-3 or d["parties"]["engine"].get("relu_flips_used", 0) > 0, d
