"""Training systems with the reference's registry and hook names (lightning/systems/__init__.py:5-14,
system.py:26-112, base_adaptor.py:21-124, meta.py:17-97, baseline.py:15-53), minus PyTorch-Lightning:
the Trainer's job on this path — one process per GPU, outer-gradient mean over ranks, clip, Adam,
Noam schedule, checkpoint cadence — is the ~60-line :class:`Trainer` below over torch.distributed
(backend "nccl" == RCCL over xGMI on MI355X, "gloo" in the CPU tests).

Everything numerical (forward, backward, inner SGD, fast weights, outer reduction over the local
tasks, clip + Adam) runs inside libmtts; the only tensor this file touches is the flat outer-gradient
buffer handed to ``all_reduce``.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Sequence

import numpy as np

from .engine import LOSS_NAMES
from .model import FastSpeech2, FastSpeech2Loss, loss2dict


def noam_lr(step: int, d_model: int, train_config) -> float:
    """lightning/optimizer.py:7 (init_lr = d_model^-0.5) x lightning/scheduler.py:11-23 (LambdaLR factor);
    ``step`` is the 0-based scheduler step."""
    o = train_config["optimizer"]
    cur = step + 1
    lr = min(cur ** -0.5, o["warm_up_step"] ** -1.5 * cur)
    for s in o["anneal_steps"]:
        if cur > s:
            lr *= o["anneal_rate"]
    return float(d_model ** -0.5 * lr)


class System:
    """lightning/systems/system.py:26 — owns the model, the loss and the outer optimiser state."""

    def __init__(self, preprocess_config, model_config, train_config, algorithm_config, log_dir=None, result_dir=None,
                 *, max_tasks: int = 1, max_batch: int = 16, max_src_len: int = 128, max_mel_len: Optional[int] = None,
                 device: int = 0, lib_path: Optional[str] = None):
        self.preprocess_config, self.model_config = preprocess_config, model_config
        self.train_config, self.algorithm_config = train_config, algorithm_config
        self.log_dir, self.result_dir = log_dir, result_dir
        self.model = FastSpeech2(preprocess_config, model_config, algorithm_config, max_tasks=max_tasks, max_batch=max_batch,
                                 max_src_len=max_src_len, max_mel_len=max_mel_len, device=device, lib_path=lib_path)
        self.loss_func = FastSpeech2Loss(preprocess_config, model_config)
        self.engine = self.model.engine
        self.global_step = 0
        self.adam_steps = 0  # optimizer.step() calls since the moments were (re)initialised: Adam's bias-correction count
        self.world_size = 1

    # system.py:53-56
    def common_step(self, batch, batch_idx, train=True):
        self.model.train(train)
        output = self.model(*(batch[2:]))
        loss = self.loss_func(batch, output)
        return loss, output

    def training_step(self, batch, batch_idx):
        """system.py:58-64 / baseline.py:25-36 — plain multi-task gradient of one batch."""
        enc = self._trained_speaker_encoder()
        if enc is not None:
            self.model.train(True)
        losses = self.engine_plain_grad([batch])
        if enc is not None:   # back-propagate dLoss/d(speaker embedding) through the LSTM encoder (autograd does this in the reference)
            assert self.world_size == 1, "a trained speaker encoder is single-rank here (its gradients are not all-reduced)"
            enc.backward(self.engine.speaker_grad(0, int(np.shape(batch[3])[0])))
            self._enc_grads_fresh = True
        return {"loss": losses[0][0], "losses": losses[0], "_batch": batch}

    def _trained_speaker_encoder(self):
        enc = getattr(self.model, "speaker_encoder", None)
        return enc if enc is not None and getattr(self.model, "spk_mode", "") in ("encoder", "scratch_encoder") else None

    def engine_plain_grad(self, batches: Sequence[tuple], total_batches: Optional[int] = None):
        self.engine.set_batches(0, list(batches))
        scale = 1.0 / (total_batches or (len(batches) * self.world_size))
        return self.engine.plain_grad(0, scale)

    def optimizer_step(self, grad_ptr: Optional[int] = None):
        o = self.train_config["optimizer"]
        lr = noam_lr(self.global_step, self.model.dims.d_model, self.train_config)
        enc = self._trained_speaker_encoder()
        if enc is not None:   # clip_grad_norm_ over ALL parameters (main.py:61): the encoder's sum of squares joins the engine's norm
            if not getattr(self, "_enc_grads_fresh", False):
                raise RuntimeError("optimizer_step with a trained speaker encoder but no encoder backward since the last step "
                                   "(only System.training_step back-propagates through it): its stale gradients would be applied "
                                   "and would skew the joint clip norm")
            self._enc_grads_fresh = False
            self.engine.set_extra_grad_sumsq(enc.grad_sumsq_ptr())
        self.engine.outer_update(lr=lr, betas=tuple(o["betas"]), eps=o["eps"], weight_decay=o["weight_decay"],
                                 max_norm=o["grad_clip_thresh"], grad_ptr=grad_ptr)
        if enc is not None:
            enc.adam_step(self.engine.grad_norm_ptr(), o["grad_clip_thresh"], lr, betas=tuple(o["betas"]), eps=o["eps"], weight_decay=o["weight_decay"])
            self.engine.set_extra_grad_sumsq(None)
        self.global_step += 1
        self.adam_steps += 1
        return lr

    # checkpoint surface (system.py:115-192 loader surgery lives in checkpoint.py)
    def state_dict(self) -> Dict[str, np.ndarray]:
        sd = {f"model.{k}": v for k, v in self.model.state_dict().items()}
        for k, v in list(sd.items()):
            mod = k.split(".")[1]
            if mod in getattr(self.model, "adapt_modules", ()):
                sd["learner.module." + k[len("model."):]] = v  # same tensors, aliased (base_adaptor.py:31-35)
        return sd


class BaseAdaptorSystem(System):
    """lightning/systems/base_adaptor.py:21"""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        a = self.algorithm_config["adapt"]
        self.adaptation_lr = a["task"]["lr"]
        self.adaptation_steps = a["train"]["steps"]
        self.test_adaptation_steps = a["test"]["steps"]
        assert self.test_adaptation_steps % self.adaptation_steps == 0  # base_adaptor.py:39

    @staticmethod
    def _on_meta_batch_start(batch):
        """base_adaptor.py:126-131"""
        assert len(batch) == 1, "meta_batch_per_gpu"
        assert len(batch[0]) == 2, "sup + qry"
        assert len(batch[0][0]) == 1, "n_batch == 1"
        assert len(batch[0][0][0]) == 12, "data with 12 elements"

    def meta_learn_tasks(self, tasks: Sequence[tuple], train: bool = True, total_tasks: Optional[int] = None):
        """adapt + meta_learn (base_adaptor.py:100-124) for all local tasks in one grouped pass.
        tasks: [(sup12, qry12), ...].  Leaves sum_t dL_q,t/dtheta / total_tasks in the outer-gradient buffer;
        returns (query losses [n][6], support losses [steps][n][6]).
        MAML order: the reference hard-wires `first_order = not train` (base_adaptor.py:107), i.e. second-order in training;
        `adapt.first_order: true|false` in the algorithm config (an extension key, absent in the reference's YAMLs) overrides
        it — BASELINE config C3 is the first-order variant."""
        fo = self.algorithm_config["adapt"].get("first_order")
        second_order = bool(train) if fo is None else (not fo)
        sup = [t[0] for t in tasks]
        qry = [t[1] for t in tasks]
        self.engine.set_batches(0, sup)
        self.engine.set_batches(1, qry, spk_from=sup, average_spk=True)
        scale = 1.0 / (total_tasks or (len(tasks) * self.world_size))
        steps = min(self.adaptation_steps, self.test_adaptation_steps)
        return self.engine.meta_grad(steps, self.adaptation_lr, scale, second_order=second_order)


    # -- few-shot test loop (base_adaptor.py:136-189) ---------------------------------------------
    def _forward_learner(self, sup_batch, qry_batch, use_fast: bool, train: bool, teacher_forced: bool):
        """forward_learner(learner, sup_batch[2], *qry_batch[3:] or [3:6], average_spk_emb=True)"""
        self.model.train(train)
        q = tuple(qry_batch) if teacher_forced else tuple(qry_batch[:6])
        args = q[2:] if teacher_forced else (q[2], q[3], q[4], q[5])
        return self.model.forward(*args, slot=1, use_fast=use_fast, spk_from=sup_batch, average_spk_emb=True)

    def _test_step(self, batch, batch_idx):
        """Step 0 with the un-adapted weights in eval mode, then cumulative first-order adaptation in chunks of
        `adapt.train.steps` up to `adapt.test.steps`; reconstruction (teacher-forced) after every chunk, free-running
        synthesis at `saving_steps`.  As in the reference the adapted clone stays in train mode."""
        outputs = {}
        test_cfg = self.algorithm_config["adapt"]["test"]
        saving_steps = test_cfg.get("saving_steps", [5, 10, 20, 50, 100])
        sup_batch, qry_batch = batch[0][0][0], batch[0][1][0]
        outputs["_batch"] = qry_batch
        preds = self._forward_learner(sup_batch, qry_batch, use_fast=False, train=False, teacher_forced=True)
        outputs["step_0"] = {"recon": {"losses": self.loss_func(qry_batch, preds), "output": preds}}
        outputs["step_0"]["synth"] = {"output": self._forward_learner(sup_batch, qry_batch, False, False, teacher_forced=False)}
        self.engine.set_batches(0, [sup_batch])
        first = True
        for ft_step in range(self.adaptation_steps, self.test_adaptation_steps + 1, self.adaptation_steps):
            self.engine.adapt(self.adaptation_steps, self.adaptation_lr, reset=first, fetch_losses=False)
            first = False
            preds = self._forward_learner(sup_batch, qry_batch, use_fast=True, train=True, teacher_forced=True)
            outputs[f"step_{ft_step}"] = {"recon": {"losses": self.loss_func(qry_batch, preds), "output": preds}}
            if ft_step in saving_steps:
                outputs[f"step_{ft_step}"]["synth"] = {"output": self._forward_learner(sup_batch, qry_batch, True, True, False)}
        return outputs

    def on_test_start(self):
        """system.py:194-212: with `adapt.test.avg_train_spk_emb` (LibriTTS, table embedding) the last 39 speaker rows — the
        unseen test speakers — start from the mean of the first 247 (train-clean-100) rows."""
        a = self.algorithm_config["adapt"]
        if a["speaker_emb"] != "table":
            return
        if self.preprocess_config["dataset"] == "LibriTTS" and a["test"].get("avg_train_spk_emb", False):
            w = self.engine.export("speaker_emb.model.weight")
            w[-39:] = w[:247].mean(axis=0)
            self.engine.load_params({"speaker_emb.model.weight": w}, strict=False)

    def test_step(self, batch, batch_idx):
        self._on_meta_batch_start(batch)
        if self.algorithm_config["adapt"]["test"].get("1-shot", False):
            # base_adaptor.py:139-147: adapt on every single support utterance in turn, same query set
            from .data import Task
            qry_batch = batch[0][1][0]
            return [self._test_step([([sup_batch], [qry_batch])], batch_idx)
                    for sup_batch in Task(sup_data=batch[0][0][0], qry_data=qry_batch, batch_size=1, shuffle=False)]
        return [self._test_step(batch, batch_idx)]


class MetaSystem(BaseAdaptorSystem):
    """lightning/systems/meta.py:17"""

    def training_step(self, batch, batch_idx):
        self._on_meta_batch_start(batch)
        q, s = self.meta_learn_tasks([(batch[0][0][0], batch[0][1][0])])
        logs = {f"Train/{k}": float(v) for k, v in zip(LOSS_NAMES, q[0])}
        return {"loss": float(q[0][0]), "losses": q[0], "log": logs, "_batch": batch[0][1][0]}

    def validation_step(self, batch, batch_idx):
        self._on_meta_batch_start(batch)
        q, s = self.meta_learn_tasks([(batch[0][0][0], batch[0][1][0])], train=False)
        return {"losses": q[0], "log": {f"Val/{k}": float(v) for k, v in zip(LOSS_NAMES, q[0])}}


class BaselineSystem(BaseAdaptorSystem):
    """lightning/systems/baseline.py:15"""

    def training_step(self, batch, batch_idx):
        assert len(batch) == 12, "data with 12 elements"
        return System.training_step(self, batch, batch_idx)

    def validation_step(self, batch, batch_idx):
        return MetaSystem.validation_step(self, batch, batch_idx)


class IMAMLSystem(BaseAdaptorSystem):
    """lightning/systems/imaml.py:22 — implicit MAML: a proximal first-order inner loop over support mini-batches
    (`adapt.imaml.batch_size`, imaml.py:50-73) and a conjugate-gradient hypergradient (`adapt.imaml.K` iterations, a fresh
    mini-batch per Hessian-vector product when `adapt.imaml.stochastic`, imaml.py:76-139; utils.py:120-189), clipped per task
    before the mean over ranks.  Not restated: the codebook phoneme-table regeneration of `on_after_batch_transfer`
    (imaml.py:27-39, `adapt.type: lang` — the front-end is outside the hot path; FastSpeech2 refuses that config)."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        im = self.algorithm_config["adapt"]["imaml"]
        self.cg_steps, self.reg_param = int(im["K"]), float(im["reg_param"])
        self.im_batch_size, self.stochastic = int(im["batch_size"]), bool(im["stochastic"])
        self.task_seed = 0

    def _task(self, sup_batch, qry_batch):
        from .data import Task
        self.task_seed += 1
        return Task(sup_data=sup_batch, qry_data=qry_batch, batch_size=self.im_batch_size, seed=self.task_seed)

    def adapt_task(self, sup_batch, qry_batch, steps: int, task=None, reset: bool = True):
        """imaml.py:50-73: `steps` regularised first-order inner steps, each on the task's next support mini-batch."""
        task = task or self._task(sup_batch, qry_batch)
        self.engine.set_inner_prox(self.reg_param)
        try:
            for k in range(steps):
                self.engine.set_batches(0, [task.next_batch()])
                self.engine.adapt(1, self.adaptation_lr, reset=(reset and k == 0), fetch_losses=False)
        finally:
            self.engine.set_inner_prox(0.0)
        return task

    def meta_learn(self, batch, batch_idx, train: bool = True, total_tasks: Optional[int] = None):
        """imaml.py:76-139.  Leaves the (per-task clipped) hypergradient / total_tasks in the outer-gradient buffer when training;
        returns the query 6-tuple."""
        sup_batch, qry_batch = batch[0][0][0], batch[0][1][0]
        task = self.adapt_task(sup_batch, qry_batch, self.adaptation_steps)
        self.engine.set_batches(0, [sup_batch])
        self.engine.set_batches(1, [qry_batch], spk_from=[sup_batch], average_spk=True)
        if not train:
            self.engine.forward(1, use_fast=True, train=True)
            return self.engine.loss(1)[0]
        q = self.engine.imaml_begin()
        task.reset_iterator()
        for _ in range(self.cg_steps):
            if self.stochastic:
                self.engine.set_batches(0, [task.next_batch()])
            self.engine.imaml_cg_step(self.adaptation_lr, self.reg_param, 1e-10)
        scale = 1.0 / (total_tasks or self.world_size)
        self.engine.imaml_finish(self.adaptation_lr, self.reg_param, scale, self.train_config["optimizer"]["grad_clip_thresh"])
        return q[0]

    def optimizer_step(self, grad_ptr: Optional[int] = None):
        """Manual optimisation (imaml.py:25,133-139): the clip already happened per task, Adam sees the reduced gradient as is."""
        o = self.train_config["optimizer"]
        lr = noam_lr(self.global_step, self.model.dims.d_model, self.train_config)
        self.engine.outer_update(lr=lr, betas=tuple(o["betas"]), eps=o["eps"], weight_decay=o["weight_decay"], max_norm=0.0, grad_ptr=grad_ptr)
        self.global_step += 1
        self.adam_steps += 1
        return lr

    def training_step(self, batch, batch_idx):
        self._on_meta_batch_start(batch)
        q = self.meta_learn(batch, batch_idx, train=True)
        return {"loss": float(q[0]), "losses": q, "log": {f"Train/{k}": float(v) for k, v in zip(LOSS_NAMES, q)}, "_batch": batch[0][1][0]}

    def validation_step(self, batch, batch_idx):
        self._on_meta_batch_start(batch)
        q = self.meta_learn(batch, batch_idx, train=False)
        return {"losses": q, "log": {f"Val/{k}": float(v) for k, v in zip(LOSS_NAMES, q)}}

    def _test_step(self, batch, batch_idx):
        """imaml.py:163-195: as BaseAdaptorSystem._test_step but every step-0 pass runs the learner in its current (train) mode
        and the adaptation is the regularised mini-batch loop continued on one Task."""
        outputs = {}
        sup_batch, qry_batch = batch[0][0][0], batch[0][1][0]
        outputs["_batch"] = qry_batch
        preds = self._forward_learner(sup_batch, qry_batch, use_fast=False, train=True, teacher_forced=True)
        outputs["step_0"] = {"recon": {"losses": self.loss_func(qry_batch, preds), "output": preds}}
        outputs["step_0"]["synth"] = {"output": self._forward_learner(sup_batch, qry_batch, False, True, teacher_forced=False)}
        task = None
        for ft_step in range(self.adaptation_steps, self.test_adaptation_steps + 1, self.adaptation_steps):
            task = self.adapt_task(sup_batch, qry_batch, self.adaptation_steps, task=task, reset=(task is None))
            preds = self._forward_learner(sup_batch, qry_batch, use_fast=True, train=True, teacher_forced=True)
            outputs[f"step_{ft_step}"] = {"recon": {"losses": self.loss_func(qry_batch, preds), "output": preds}}
            if ft_step in [5, 10, 20, 50, 100]:
                outputs[f"step_{ft_step}"]["synth"] = {"output": self._forward_learner(sup_batch, qry_batch, True, True, False)}
        return outputs


SYSTEM = {"meta": MetaSystem, "imaml": IMAMLSystem, "baseline": BaselineSystem}


def get_system(algorithm: str):
    """lightning/systems/__init__.py:5-14."""
    if algorithm not in SYSTEM:
        raise KeyError(f"system type {algorithm!r} is not on the hot path (supported: {sorted(SYSTEM)})")
    return SYSTEM[algorithm]


class Trainer:
    """The slice of pl.Trainer(strategy='ddp', gradient_clip_val=...) this path needs (main.py:30-38,57-64):
    one process per GPU, every rank runs its share of the meta-batch, the flat outer gradient is summed over
    ranks (each rank already scaled by 1/total_tasks => mean, as DDP does), then every rank applies the same
    clip + Adam step."""

    def __init__(self, system: System, outer_grad_tensor=None, process_group=None):
        import torch.distributed as dist
        self.system = system
        self.dist = dist if (dist.is_available() and dist.is_initialized()) else None
        self.group = process_group
        self.system.world_size = self.dist.get_world_size(self.group) if self.dist else 1
        self.outer = outer_grad_tensor  # torch view of engine.outer_grad_ptr() (zero copy on GPU)
        if outer_grad_tensor is not None and int(outer_grad_tensor.numel()) != int(system.engine.sync_floats):
            # the collective covers the gradient AND the exchange tail behind it (loss scalars + BatchNorm buffers): a view of the
            # gradient alone would leave every non-zero rank unpacking its own zeroed tail into its BatchNorm running buffers
            raise ValueError(f"outer_grad_tensor has {int(outer_grad_tensor.numel())} elements; the exchange buffer is engine.sync_floats = "
                             f"{int(system.engine.sync_floats)} (outer gradient + exchange tail): build it from engine.outer_grad_view()")
        self.library_comm = False
        # accumulate_grad_batches (main.py:62; config/train/base.yaml: optimizer.grad_acc_step): gradients of N consecutive batches are
        # summed (each scaled by 1/N, as PL divides the loss), the ranks reduce and the optimizer steps on the N-th — DDP's no_sync
        # behaviour: no collective on the accumulating batches
        self.grad_acc = int(system.train_config["optimizer"].get("grad_acc_step", 1))
        if self.grad_acc < 1:
            raise ValueError(f"optimizer.grad_acc_step must be >= 1, got {self.grad_acc}")
        self._acc_i = 0
        if self.dist is not None and self.dist.get_backend(self.group) == "nccl":
            # RCCL orders its collective after torch's CURRENT stream of the engine's device: put the engine's launches on that
            # stream so the all-reduce sees the finished outer gradient and the clip + Adam after it sees the reduced one
            import torch
            dev = self.system.engine.device
            self.system.engine.set_stream(torch.cuda.current_stream(dev).cuda_stream)
            enc = self.system._trained_speaker_encoder()
            if enc is not None:   # the encoder and the engine exchange device scalars (joint clip norm): same stream, same order
                enc.set_stream(torch.cuda.current_stream(dev).cuda_stream)
            if outer_grad_tensor is None and self.system.world_size > 1:
                # the all-reduce itself runs inside the library (mtts_allreduce_outer); torch only carries the unique id
                # lock-step bring-up (every rank probes the library, the probes are MIN-reduced, only then is rank 0's id broadcast and
                # ncclCommInitRank entered): a rank whose librccl cannot be loaded must not leave the others waiting in the init
                eng = self.system.engine
                rank = self.dist.get_rank(self.group)
                try:
                    ok = eng.comm_available()            # every rank: can librccl be loaded here?
                    uid = eng.comm_unique_id() if (ok and rank == 0) else None   # the id itself is rank 0's alone
                    ok = ok and (rank != 0 or uid is not None)
                except Exception:  # noqa: BLE001
                    ok, uid = False, None
                flag = torch.tensor([1 if ok else 0], device=f"cuda:{dev}")
                self.dist.all_reduce(flag, op=self.dist.ReduceOp.MIN, group=self.group)
                if bool(flag.item()):
                    ids = [uid]
                    self.dist.broadcast_object_list(ids, src=self.dist.get_global_rank(self.group, 0) if self.group is not None else 0, group=self.group)
                    try:
                        eng.comm_init(ids[0], rank, self.system.world_size)
                        ok = True
                    except Exception:  # noqa: BLE001
                        ok = False
                    # a rank whose ncclCommInitRank failed must not leave the others blocked in the first ncclAllReduce: agree on the outcome
                    flag = torch.tensor([1 if ok else 0], device=f"cuda:{dev}")
                    self.dist.all_reduce(flag, op=self.dist.ReduceOp.MIN, group=self.group)
                    self.library_comm = bool(flag.item())   # False: torch.distributed's all_reduce on the zero-copy view (_allreduce)

    def _allreduce(self):
        """The exchange step: SUM of the flat outer gradient AND of the tail behind it — the six loss scalars (log_dict(sync_dist=True),
        meta.py:78-79) and the PostNet BatchNorm running buffers (DDP broadcast_buffers, main.py:32: every rank continues with rank 0's)."""
        eng = self.system.engine
        if self.dist is None or self.system.world_size == 1:
            eng.sync_pack(1.0)          # (one rank: the "reduced" losses are its own)
            return
        if self.library_comm:
            eng.allreduce_outer()       # packs, reduces and unpacks inside the library
            return
        if self.outer is None:
            import torch
            self.outer = torch.as_tensor(eng.outer_grad_view(), device=f"cuda:{eng.device}")
        rank = self.dist.get_rank(self.group)
        eng.sync_pack(eng.bn_pack_weight(rank, self.system.world_size))   # (the engine's BatchNorm sync mode, as the library path)
        self.dist.all_reduce(self.outer, op=self.dist.ReduceOp.SUM, group=self.group)
        eng.sync_unpack()

    def synced_losses(self):
        """Mean over ALL tasks (all ranks) of the six losses of the last gradient call — what the reference logs with `sync_dist=True` at that
        `training_step` (meta.py:78-79).  With gradient accumulation (grad_acc_step = N) that is the window's LAST micro-batch, as in the
        reference, where every micro-batch logs its own losses; the exchange tail carries the sums scaled by grad_scale = 1 / (tasks x N) — the
        accumulation divisor is undone here."""
        return self.system.engine.synced_losses() * float(self.grad_acc)

    def _with_accumulation(self, grad_call):
        """Runs one gradient call of an accumulation window.  The engine's accumulate flag is armed for THIS call only (cleared in a
        finally: a later direct System.training_step / engine.meta_grad must overwrite the outer buffer, not add to a stale one) and
        the window advances only when the call succeeded (an exception mid-window leaves the window where it was).  Returns
        (result, hold): hold = the optimizer must NOT step yet."""
        eng = self.system.engine
        eng.set_grad_accumulation(self._acc_i > 0)
        if self.library_comm and self._acc_i == self.grad_acc - 1:
            # the window's last gradient call fills the buffer the ranks exchange: its buckets leave on the communication stream as the
            # backward completes them (DDP's bucketed all-reduce overlapping the backward, main.py:30-38); _allreduce then only joins them
            eng.arm_allreduce_overlap()
        try:
            out = grad_call()
        finally:
            eng.set_grad_accumulation(False)
            eng.disarm_allreduce_overlap()      # (a Python exception in front of the engine call: the arming must not reach a later, unrelated gradient call)
        self._acc_i += 1
        if self._acc_i < self.grad_acc:
            return out, True
        self._acc_i = 0
        return out, False

    def meta_step(self, local_tasks: Sequence[tuple], total_tasks: int):
        """One batch.  With grad_acc_step = N the optimizer steps on every N-th call (returned lr is None in between)."""
        (q, s), hold = self._with_accumulation(lambda: self.system.meta_learn_tasks(local_tasks, total_tasks=total_tasks * self.grad_acc))
        if hold:
            return q, s, None
        self._allreduce()
        lr = self.system.optimizer_step()
        return q, s, lr

    def imaml_step(self, batch, total_tasks: int):
        """IMAMLSystem.meta_learn(train=True) on this rank's task, mean over ranks (imaml.py:132 `reduce`), manual optimizer step."""
        if self.grad_acc != 1:
            raise NotImplementedError("optimizer.grad_acc_step != 1 with the iMAML system (its hypergradient overwrites the outer buffer per task)")
        q = self.system.meta_learn(batch, 0, train=True, total_tasks=total_tasks)
        self._allreduce()
        lr = self.system.optimizer_step()
        return q, lr

    def plain_step(self, local_batches: Sequence[tuple], total_batches: int):
        if self.system._trained_speaker_encoder() is not None:
            # the LSTM speaker encoder's backward lives in System.training_step (one batch, one rank): stepping the optimizer from here
            # would update it from stale gradients and skew the joint clip norm
            raise NotImplementedError("speaker_emb: encoder / scratch_encoder train through System.training_step (single batch, single rank)")
        losses, hold = self._with_accumulation(lambda: self.system.engine_plain_grad(local_batches, total_batches * self.grad_acc))
        if hold:
            return losses, None
        self._allreduce()
        lr = self.system.optimizer_step()
        return losses, lr
