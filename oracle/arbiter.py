"""ORACLE — test infrastructure, not product code.

fp64 arbiter of gradient parity.  Two correct fp32 implementations of this piecewise-smooth loss
can disagree on a gradient tensor by per cents although every forward value agrees to 1e-5: the two mel L1 terms (lightning/model/loss.py:
d|x|/dx = sign(x)) and every ReLU (transformer/SubLayers.py:89, lightning/model/modules.py:217-229) switch a unit's WHOLE contribution at a
point that sits inside fp32 noise for a handful of units per pass.  Comparing the engine with the fp32 oracle cannot tell such a flip from a
bug.  The arbiter therefore evaluates the same task in float64 (same weights, same dropout masks, oracle/fs2_oracle.py run in double) and, per
sampled tensor, reports for each fp32 party X in {engine, oracle32}

    raw        max |g_X - g64| / max |g64|                     — what a plain fp64 comparison sees
    l1         the same after the L1 signs of the AMBIGUOUS mel / mel_post elements (|prediction - target| < 1e-4 in fp64) were taken from
               X's OWN forward output (its exported mel / mel_post): an exact correction — one extra backward of
               sum_{flipped} (s_X - s_64) * prediction / N through the fp64 graph
    explained  (only for tensors that still fail) the same after the contributions of single ReLU units whose fp64 pre-activation is within 1e-6
               of zero were switched where that REDUCES the residual: unit u contributes c_u = dL/dy_u * d pre_u / d theta (one backward from the
               pre-activation with a one-hot seed), fp64 has it on or off, X may have decided the other way — each flip is an identified unit
               with an exactly priced contribution, not a tolerance.

Gate (per tensor):  err(engine) <= 3 * err(oracle32) + 1e-3, with err = `l1`, or `explained` where computed; relative to max |g64|.
A dense 1-3 % gradient bug cannot be bought off: an L1 flip must be visible in X's own forward output, and at most the ~20 listed ReLU units
(of ~1e7 per pass) may be switched.  Nothing here is timed or shipped.
"""
from __future__ import annotations

from typing import Dict, Optional, Sequence

import numpy as np
import torch

from . import fs2_oracle as O

GATE_FACTOR = 3.0
GATE_FLOOR = 1e-3
L1_EPS = 1e-4
RELU_EPS = 1e-6          # reported: units this close to zero in float64 (the count the fp32 oracle's KINK_LOG uses)
RELU_BAND = 1e-4         # candidates for a flip: an fp32 forward after five inner steps agrees with float64 to ~2e-5 (mel), so a unit whose float64
                         # pre-activation is inside this band (5x that) can sit on the other side of zero in a correct fp32 implementation
MAX_PRICED = 64          # candidates priced exactly per party (the best by the screening score <residual, contribution>)
MAX_FLIPS = 32           # flips a party may be granted per task: "a handful" of ~1e7 units


def f64_params(np_params: Dict[str, np.ndarray]):
    p = {k: torch.from_numpy(np.asarray(v)).to(torch.float64) for k, v in np_params.items()}
    for k, v in p.items():
        if not k.endswith(("position_enc", "pitch_bins", "energy_bins")):
            v.requires_grad_(True)
    return p


def f64_batch(batch):
    out = []
    for i, x in enumerate(batch):
        if isinstance(x, np.ndarray):
            t = torch.from_numpy(x)
            out.append(t.to(torch.float64) if t.is_floating_point() else t)
        else:
            out.append(int(x) if i in (5, 8) else x)
    return tuple(out)


def _rel(a: torch.Tensor, ref_max: float) -> float:
    return float(a.abs().max()) / max(ref_max, 1e-300)


def arbitrate_task(np_params, np_buffers, sup, qry, *, modules: Sequence[str], n_head, max_seq_len: int, steps: int, lr: float, masks,
                   names: Sequence[str], parties: Dict[str, dict], second_order: bool = False, explain: bool = True) -> dict:
    """One task of a meta-step in float64 and the verdict on each party's sampled gradient tensors.

    parties[X] = {"grads": {name: ndarray (the per-task query gradient, unscaled)}, "mel": (B, T, n_mel), "mel_post": (B, T, n_mel)} — X's own
    query-pass outputs (padded frames are ignored).  `masks`: the steps + 1 DropoutMasks of the task, or None.  Returns a JSON-able report."""
    p = f64_params(np_params)
    buf = {k: torch.from_numpy(np.asarray(v).copy()).to(torch.float64) if np.asarray(v).dtype.kind == "f" else torch.from_numpy(np.asarray(v).copy())
           for k, v in np_buffers.items()}
    tb_s, tb_q = f64_batch(sup), f64_batch(qry)
    taps = []
    # inner loop + query pass in double; taps only for the query pass (a flipped unit of an inner step moves the fast weights by lr * c_u)
    anames = O.adapted_names(p, modules)
    fast = {k: p[k] for k in anames}
    for s_ in range(steps):
        cur = dict(p); cur.update(fast)
        preds = O.fs2_forward(cur, buf, *tb_s[2:], n_head=n_head, max_seq_len=max_seq_len, training=True, dropout=masks[s_] if masks else None)
        loss = O.fs2_loss(tb_s, preds)
        gr = torch.autograd.grad(loss[0], [fast[k] for k in anames], create_graph=second_order)
        fast = {k: fast[k] - lr * g for k, g in zip(anames, gr)}
    cur = dict(p); cur.update(fast)
    O.RELU_TAPS = taps
    try:
        preds = O.fs2_forward(cur, buf, tb_s[2], *tb_q[3:], n_head=n_head, max_seq_len=max_seq_len, training=True, average_spk_emb=True,
                              dropout=masks[steps] if masks else None)
    finally:
        O.RELU_TAPS = None
    ql = O.fs2_loss(tb_q, preds)
    mel, mel_post, mel_masks = preds[0], preds[1], preds[7]
    valid = (~mel_masks).unsqueeze(-1)                                   # (B, T, 1)
    target = tb_q[6][:, : mel_masks.shape[1], :]
    n_valid = float(valid.sum()) * mel.shape[-1]
    theta = [p[n] for n in names]
    grads = torch.autograd.grad(ql[0], theta, retain_graph=True, allow_unused=True)
    g64 = {n: (g if g is not None else torch.zeros_like(p[n])) for n, g in zip(names, grads)}
    gmax = {n: float(g64[n].abs().max()) for n in names}
    floor = 1e-4 * max(gmax.values())      # a tensor whose gradient vanishes in exact arithmetic (w_ks.bias: softmax shift invariance) is pure roundoff
    gmax = {n: max(v, floor) for n, v in gmax.items()}
    # ---- L1 sign transfer ------------------------------------------------------------------------------------------------------------
    res = {"mel": (mel - target).detach(), "mel_post": (mel_post - target).detach()}
    amb = {k: (v.abs() < L1_EPS) & valid for k, v in res.items()}
    report = {"query_losses_f64": [float(x) for x in ql], "l1_ambiguous_elements": int(sum(int(a.sum()) for a in amb.values())),
              "relu_ambiguous_units": int(sum(int((x.detach().abs() < RELU_EPS).sum()) for x, _ in taps)), "parties": {}, "tensors": {}}
    adj = {}
    for X, d in parties.items():
        corr_loss, flips = 0.0, 0
        for key, pred in (("mel", mel), ("mel_post", mel_post)):
            out_x = torch.from_numpy(np.asarray(d[key], np.float64))[:, : target.shape[1], :]
            s_x = torch.sign(out_x - target)
            s_64 = torch.sign(res[key])
            flip = amb[key] & (s_x != s_64)
            flips += int(flip.sum())
            if bool(flip.any()):
                corr_loss = corr_loss + (((s_x - s_64) * flip) * pred).sum() / n_valid
            # outside the ambiguous band the signs must agree: a forward-value disagreement, not a kink
            hard = valid & ~amb[key] & (s_x != s_64)
            report["parties"].setdefault(X, {})[f"{key}_sign_disagreements_outside_band"] = int(hard.sum())
        report["parties"][X]["l1_flips"] = flips
        if flips:
            cg = torch.autograd.grad(corr_loss, theta, retain_graph=True, allow_unused=True)
            adj[X] = {n: g64[n] + (c if c is not None else 0.0) for n, c in zip(names, cg)}
        else:
            adj[X] = g64
    for n in names:
        row = {}
        for X, d in parties.items():
            gx = torch.from_numpy(np.asarray(d["grads"][n], np.float64))
            row[X] = {"raw": _rel(gx - g64[n], gmax[n]), "l1": _rel(gx - adj[X][n], gmax[n])}
        report["tensors"][n] = row

    def gate(row):
        e = row["engine"].get("explained", row["engine"]["l1"])
        o = row["oracle32"].get("explained", row["oracle32"]["l1"]) if "oracle32" in row else 0.0
        return e <= GATE_FACTOR * o + GATE_FLOOR

    # ---- ReLU units: only for tensors that still fail -----------------------------------------------------------------------------------
    failing = [n for n in names if not gate(report["tensors"][n])]
    if failing and explain:
        ftheta = [p[n] for n in failing]
        # candidates: units of the query pass whose float64 pre-activation lies inside the band an fp32 forward can put on the other side of zero
        # (forward values of two fp32 implementations agree to ~2e-5 after five inner steps) and that something downstream listens to
        all_g = torch.autograd.grad(ql[0], [y for _, y in taps], retain_graph=True, allow_unused=True)
        cand = []     # (tap index, flat index, direction, dL/dy_u)
        for ti, ((x, y), gy) in enumerate(zip(taps, all_g)):
            if gy is None:
                continue
            m = (x.detach().abs() < RELU_BAND) & (gy != 0)
            for fi in m.reshape(-1).nonzero().reshape(-1).tolist():
                cand.append((ti, fi, -1.0 if float(x.reshape(-1)[fi]) > 0 else 1.0, float(gy.reshape(-1)[fi])))
        report["relu_candidates_in_band"] = len(cand)

        def tap_values(params_over):
            t2 = []
            O.RELU_TAPS = t2
            try:
                with torch.no_grad():
                    c2 = dict(cur); c2.update(params_over)
                    O.fs2_forward(c2, {k: v.clone() for k, v in buf.items()}, tb_s[2], *tb_q[3:], n_head=n_head, max_seq_len=max_seq_len, training=True,
                                  average_spk_emb=True, dropout=masks[steps] if masks else None)
            finally:
                O.RELU_TAPS = None
            return [x.detach().reshape(-1) for x, _ in t2]

        for X, d in parties.items():
            r = {n: torch.from_numpy(np.asarray(d["grads"][n], np.float64)) - adj[X][n] for n in failing}
            if max(_rel(r[n], gmax[n]) for n in failing) <= GATE_FLOOR or not cand:
                report["parties"][X]["relu_flips_used"] = 0
                continue
            # screening: <r, c_u> = dL/dy_u * (J r)_u for EVERY candidate from ONE directional derivative of the pre-activations along r
            # (central difference in float64 — first-order MAML: the query pass sees theta through fast = theta - lr * sum g, d fast / d theta = I);
            # the best candidates are priced exactly (one backward each), switched greedily where that shrinks the residual's L2 norm, and the
            # screening is repeated on the new residual (contributions of neighbouring units overlap, so a flipped unit can hide behind another)
            units, priced, taken = [], set(), []
            for _round in range(4):
                rn = float(torch.sqrt(sum((r[n] ** 2).sum() for n in failing)))
                if rn == 0.0 or len(units) >= MAX_PRICED:
                    break
                h = 1e-4
                plus = tap_values({n: cur[n].detach() + (h / rn) * r[n] for n in failing})
                minus = tap_values({n: cur[n].detach() - (h / rn) * r[n] for n in failing})
                scored = []
                for ci, (ti, fi, sgn, gu) in enumerate(cand):
                    if ci in priced:
                        continue
                    jr = float(plus[ti][fi] - minus[ti][fi]) / (2.0 * h) * rn
                    scored.append((sgn * gu * jr, ci))
                scored.sort(reverse=True)
                fresh = 0
                for sc, ci in scored[: max(0, min(16, MAX_PRICED - len(units)))]:
                    if sc <= 0:
                        break
                    ti, fi, sgn, gu = cand[ci]
                    x = taps[ti][0]
                    seed = torch.zeros_like(x).reshape(-1)
                    seed[fi] = gu
                    cu = torch.autograd.grad(x, ftheta, grad_outputs=seed.reshape(x.shape), retain_graph=True, allow_unused=True)
                    units.append((sgn, {n: (c if c is not None else torch.zeros_like(p[n])) for n, c in zip(failing, cu)}, float(x.reshape(-1)[fi])))
                    priced.add(ci)
                    fresh += 1
                before = len(taken)
                for _sweep in range(2):
                    for u, (sgn, cu, xv) in enumerate(units):
                        if u in taken or len(taken) >= MAX_FLIPS:
                            continue
                        dot = sgn * sum(float((r[n] * cu[n]).sum()) for n in failing)
                        nrm = sum(float((cu[n] ** 2).sum()) for n in failing)
                        if nrm > 0 and 2.0 * dot > nrm:
                            for n in failing:
                                r[n] = r[n] - sgn * cu[n]
                            taken.append(u)
                if fresh == 0 and len(taken) == before:
                    break
            for n in failing:
                report["tensors"][n][X]["explained"] = _rel(r[n], gmax[n])
            report["parties"][X]["relu_flips_used"] = len(taken)
            report["parties"][X]["relu_units_priced"] = len(units)
            report["parties"][X]["relu_flipped_preactivations_f64"] = [units[u][2] for u in taken]
    for n in names:
        report["tensors"][n]["ok"] = bool(gate(report["tensors"][n]))
    report["pass"] = all(report["tensors"][n]["ok"] for n in names)
    report["gate"] = f"err(engine) <= {GATE_FACTOR:g} * err(oracle32) + {GATE_FLOOR:g} (max-norm, relative to max |g64|; L1 signs of ambiguous elements from each party's own forward)"
    return report


def summarize(report: dict) -> dict:
    """The per-task digest a bench line / a test message carries: worst tensor per error kind and party, the flips, the failing tensors."""
    out = {"pass": report["pass"], "l1_ambiguous_elements": report["l1_ambiguous_elements"], "relu_ambiguous_units": report["relu_ambiguous_units"],
           "parties": report["parties"], "failing": [n for n, r in report["tensors"].items() if not r["ok"]]}
    for X in report["parties"]:
        for kind in ("raw", "l1", "explained"):
            vals = {n: r[X][kind] for n, r in report["tensors"].items() if X in r and kind in r[X]}
            if vals:
                w = max(vals, key=vals.get)
                out[f"{X}_{kind}_max"] = {"err": vals[w], "tensor": w}
        gated = {n: r[X].get("explained", r[X]["l1"]) for n, r in report["tensors"].items() if X in r}     # the figure the gate compares
        if gated:
            w = max(gated, key=gated.get)
            out[f"{X}_gated_max"] = {"err": gated[w], "tensor": w}
    return out


def synth_task_worker(job: dict) -> dict:
    """Process-pool entry (bench.py parity_check, the `-m "gpu and slow"` test): task `j` of the synthetic meta-batch rebuilt from its seeds
    (meta_tts_amd.synth) in float64 with the engine's dropout masks, judged against the parties' data in `job`."""
    import os
    torch.set_num_threads(int(job.get("threads", 8)))
    from meta_tts_amd import synth
    from meta_tts_amd.config import ModelDims
    from .dropout_masks import DropoutMasks, plan_seed
    j = int(job["task"])
    if "model" in job:         # (bench.py --selftest-emu: a tiny model and its batches, handed over instead of rebuilt from the task's seeds)
        mc, pc, nspk, vocab = job["model"]
        dims = ModelDims(mc, pc, n_speaker=nspk, vocab=vocab)
        sup, qry = job["sup"], job["qry"]
    else:
        dims = ModelDims()
        sup, qry = synth.make_task(j)
    masks = None
    if job.get("dropout_seed") is not None:
        probs = dict(enc=dims.enc_dropout, dec=dims.dec_dropout, vp=dims.vp_dropout, postnet=0.5)
        masks = [DropoutMasks(plan_seed(int(job["dropout_seed"]), k + 1), int(job.get("group_index", j)), probs) for k in range(int(job["steps"]) + 1)]
    rep = arbitrate_task(synth.make_params(dims, 0, weight_scale=float(job["weight_scale"])), synth.make_buffers(dims), sup, qry, modules=job["modules"],
                         n_head=(dims.enc_heads, dims.dec_heads), max_seq_len=dims.max_seq_len, steps=int(job["steps"]), lr=float(job["lr"]), masks=masks,
                         names=list(job["names"]), parties=job["parties"], second_order=bool(job.get("second_order", False)), explain=bool(job.get("explain", True)))
    rep["task"] = j
    return rep


def run_pool(jobs, processes: int, timeout_s: float = 900.0):
    """The jobs on `processes` spawned workers (fresh interpreters: no fork of a process that holds a GPU context)."""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    with ctx.Pool(processes) as pool:
        return pool.map_async(synth_task_worker, jobs).get(timeout_s)
