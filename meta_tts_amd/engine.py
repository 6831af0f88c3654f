"""Thin Python handle over libmtts (include/mtts.h): owns nothing but the handle pointer; all device
memory and all compute live behind the C ABI."""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence

import numpy as np

from . import _lib
from .config import ModelDims

MODULE_BITS = {"encoder": 0, "variance_adaptor": 1, "decoder": 2, "mel_linear": 3, "postnet": 4, "speaker_emb": 5}
LOSS_NAMES = ("Total Loss", "Mel Loss", "Mel-Postnet Loss", "Pitch Loss", "Energy Loss", "Duration Loss")


class MttsError(RuntimeError):
    pass


def _arr(x, dtype):
    a = np.ascontiguousarray(x, dtype=dtype)
    return a


class _DevView:
    """Zero-copy view of a device buffer for torch.as_tensor (CUDA array interface v2)."""

    def __init__(self, ptr: int, n: int):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f4", "data": (ptr, False), "version": 2}


class Engine:
    def __init__(self, dims: ModelDims, adapt_modules: Sequence[str] = (), max_tasks: int = 1, max_B: int = 16,
                 max_S: int = 128, max_T: int = 1000, device: int = 0, lib_path: Optional[str] = None, shared_speaker: bool = False):
        self.lib = _lib.load(lib_path)
        self.dims = dims
        cfg = _lib.ModelCfg()
        for f in ("d_model", "enc_layers", "dec_layers", "enc_heads", "dec_heads", "d_ff", "k1", "k2", "vp_filter",
                  "vp_kernel", "n_bins", "max_seq_len", "n_mel", "vocab", "n_speaker", "postnet_dim", "postnet_kernel",
                  "postnet_layers", "pitch_min", "pitch_max", "energy_min", "energy_max", "enc_dropout", "dec_dropout", "vp_dropout"):
            setattr(cfg, f, getattr(dims, f))
        cfg.pitch_frame_level, cfg.energy_frame_level = int(dims.pitch_frame_level), int(dims.energy_frame_level)
        mask = 0
        for m in adapt_modules:
            if m not in MODULE_BITS:
                raise MttsError(f"adapt module {m!r} is not supported (supported: {sorted(MODULE_BITS)})")
            mask |= 1 << MODULE_BITS[m]
        cfg.adapt_mask = mask
        # adapt.speaker_emb == "shared" (speaker_encoder.py:52-53,67-69): a one-row table looked up with zeros_like(speaker ids)
        self.shared_speaker = bool(shared_speaker)
        self.speaker_encoder = None   # speaker_emb: dvec — callable (ref_mels, ref_slices) -> (B, d_model), e.g. speaker_encoder.DVectorEncoder
        if self.shared_speaker and dims.n_speaker != 1:
            raise MttsError("a shared speaker embedding is a table with exactly one row")
        self.adapt_modules = tuple(adapt_modules)
        self.max_tasks = max_tasks
        h = C.c_void_p()
        rc = self.lib.mtts_create(C.byref(cfg), device, max_tasks, max_B, max_S, max_T, C.byref(h))
        if rc != 0:
            raise MttsError("mtts_create: " + (self.lib.mtts_last_error(None) or b"").decode())
        self.h = h
        self._keep: List = []
        self.n_tasks = [0, 0]
        self.batch_shapes = [[], []]
        self.epoch = 0  # bumped whenever a batch plan or the (single, shared) activation workspace is replaced: stale-Predictions guard (model.py)
        self.device = device
        # parameter table
        self.params: Dict[str, tuple] = {}
        name = C.c_char_p(); ndim = C.c_int(); shape = (C.c_int * 4)(); off = C.c_int64(); ad = C.c_int()
        for i in range(self.lib.mtts_param_count(self.h)):
            self._ck(self.lib.mtts_param_info(self.h, i, C.byref(name), C.byref(ndim), C.byref(shape), C.byref(off), C.byref(ad)))
            self.params[name.value.decode()] = (tuple(shape[k] for k in range(ndim.value)), int(off.value), bool(ad.value))
        self.n_total = int(self.lib.mtts_param_total(self.h))
        self.adapt_start = int(self.lib.mtts_adapt_start(self.h))

    # ------------------------------------------------------------------------------
    def _ck(self, rc: int):
        if rc != 0:
            raise MttsError((self.lib.mtts_last_error(self.h) or b"unknown error").decode())

    def close(self):
        if getattr(self, "h", None):
            self.lib.mtts_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, stream_ptr: int):
        self._ck(self.lib.mtts_set_stream(self.h, C.c_void_p(stream_ptr)))

    def set_dropout(self, enable: bool, seed: int = 0):
        """Train-mode dropout on/off (off = parity configuration)."""
        self._ck(self.lib.mtts_set_dropout(self.h, int(enable), int(seed) & 0xFFFFFFFF))

    def reserve_second_order(self, steps: int):
        """Allocate up front what a second-order meta_grad of `steps` inner steps would allocate on first use (include/mtts.h)."""
        self._ck(self.lib.mtts_reserve_second_order(self.h, int(steps)))

    def set_numerics(self, mode: str = "fp32"):
        """Arithmetic of the contractions: "fp32" (default; the reference's own and the parity mode) or "bf16" (bf16 operands, fp32
        accumulation, the long convolutions fed from bf16 operand planes: BASELINE.json configs[1]) or "bf16-staged" (the same without
        the planes; include/mtts.h: mtts_set_numerics)."""
        modes = {"fp32": 0, "bf16": 1, "bf16-staged": 2}
        if mode not in modes:
            raise ValueError("numerics mode must be one of %s" % sorted(modes))
        self._ck(self.lib.mtts_set_numerics(self.h, modes[mode]))

    def set_grad_accumulation(self, accumulate: bool):
        """meta_grad / plain_grad add to the outer-gradient buffer instead of overwriting it (gradient accumulation, main.py:62)."""
        self._ck(self.lib.mtts_set_grad_accumulation(self.h, int(bool(accumulate))))

    def synchronize(self):
        self._ck(self.lib.mtts_synchronize(self.h))

    def profile_gemm(self, enable: bool):
        self._ck(self.lib.mtts_profile_gemm(self.h, int(enable)))

    def profile_report(self) -> Dict[str, np.ndarray]:
        """{kernel name as a rocprofv3 kernel trace prints it: [launches, ms, algorithmic flops, algorithmic bytes]} for the GEMM
        kernels this handle launched since profile_gemm(True) (include/mtts.h)."""
        kinds = int(self.lib.mtts_profile_kinds())
        rep = (C.c_double * (4 * kinds))()
        self._ck(self.lib.mtts_profile_report(self.h, rep, kinds))
        arr = np.array(list(rep), np.float64).reshape(kinds, 4)
        return {self.lib.mtts_profile_kernel_name(k).decode(): arr[k] for k in range(kinds) if arr[k, 0] > 0}

    # ---- parameters --------------------------------------------------------------
    def load_params(self, params: Dict[str, np.ndarray], strict: bool = True):
        for name, (shape, _, _) in self.params.items():
            if name not in params:
                if strict:
                    raise MttsError(f"missing parameter {name}")
                continue
            a = _arr(params[name], np.float32)
            if tuple(a.shape) != shape:
                raise MttsError(f"shape mismatch for {name}: {a.shape} vs {shape}")
            self._ck(self.lib.mtts_load_param(self.h, name.encode(), a.ctypes.data_as(C.c_void_p), a.size))

    def import_state(self, name: str, which: int, value: np.ndarray):
        a = _arr(value, np.float32)
        if tuple(a.shape) != self.params[name][0]:
            raise MttsError(f"shape mismatch for {name}")
        self._ck(self.lib.mtts_import_state(self.h, name.encode(), which, a.ctypes.data_as(C.c_void_p), a.size))

    def set_optimizer_step(self, step: int):
        self._ck(self.lib.mtts_set_optimizer_step(self.h, int(step)))

    def export(self, name: str, which: int = 0, task: int = 0) -> np.ndarray:
        shape = self.params[name][0]
        out = np.empty(shape, np.float32)
        self._ck(self.lib.mtts_export_param(self.h, name.encode(), which, task, out.ctypes.data_as(C.c_void_p), out.size))
        return out

    def state_dict(self) -> Dict[str, np.ndarray]:
        return {n: self.export(n, 0) for n in self.params}

    def grads(self, which: int = 2, task: int = 0, names: Optional[Sequence[str]] = None) -> Dict[str, np.ndarray]:
        return {n: self.export(n, which, task) for n in (names or self.params)}

    def set_bn_buffers(self, layer: int, mean: np.ndarray, var: np.ndarray, tracked: int = 0):
        m, v = _arr(mean, np.float32), _arr(var, np.float32)
        self._ck(self.lib.mtts_set_bn_buffers(self.h, layer, m.ctypes.data_as(C.c_void_p), v.ctypes.data_as(C.c_void_p), tracked))

    def get_bn_buffers(self, layer: int):
        c = self.dims.n_mel if layer == self.dims.postnet_layers - 1 else self.dims.postnet_dim
        m, v, t = np.empty(c, np.float32), np.empty(c, np.float32), C.c_int64()
        self._ck(self.lib.mtts_get_bn_buffers(self.h, layer, m.ctypes.data_as(C.c_void_p), v.ctypes.data_as(C.c_void_p), C.byref(t)))
        return m, v, int(t.value)

    def set_bins(self, pitch_bins=None, energy_bins=None):
        nb = self.dims.n_bins - 1
        pb = _arr(pitch_bins, np.float32) if pitch_bins is not None else None
        eb = _arr(energy_bins, np.float32) if energy_bins is not None else None
        for a in (pb, eb):
            if a is not None and a.shape != (nb,):
                raise MttsError(f"bins must have shape ({nb},)")
        self._ck(self.lib.mtts_set_bins(self.h, pb.ctypes.data_as(C.c_void_p) if pb is not None else None,
                                        eb.ctypes.data_as(C.c_void_p) if eb is not None else None, nb))

    def get_bins(self):
        nb = self.dims.n_bins - 1
        pb, eb = np.empty(nb, np.float32), np.empty(nb, np.float32)
        self._ck(self.lib.mtts_get_bins(self.h, pb.ctypes.data_as(C.c_void_p), eb.ctypes.data_as(C.c_void_p), nb))
        return pb, eb

    # ---- batches -----------------------------------------------------------------
    @staticmethod
    def _is_ref_mel_args(sa):
        return isinstance(sa, (tuple, list)) and len(sa) == 2 and isinstance(sa[1], (tuple, list))

    def _has_speaker_embeddings(self, b):
        sa = b[2]
        if self._is_ref_mel_args(sa):
            return True
        if hasattr(sa, "detach"):
            sa = sa.detach().cpu().numpy()
        return np.ndim(sa) == 2 and np.asarray(sa).dtype.kind == "f"

    def _embed_speaker_args(self, sa):
        if self._is_ref_mel_args(sa):
            if self.speaker_encoder is None:
                raise MttsError("batch carries (ref_mels, ref_slices) speaker args but no speaker encoder is attached (Engine.speaker_encoder)")
            return np.asarray(self.speaker_encoder(sa), np.float32)
        if hasattr(sa, "detach"):
            sa = sa.detach().cpu().numpy()
        return np.asarray(sa, np.float32)

    def _cbatch(self, b, keep):
        """12-tuple (collate.py:47-60) of numpy arrays / torch CPU tensors -> mtts_batch.  A tuple whose
        mels / durations are None (or that stops after max_src_len, i.e. ``batch[:6]``) is a free-running batch."""
        def np_(x, dt):
            if hasattr(x, "detach"):
                x = x.detach().cpu().numpy()
            a = _arr(x, dt)
            keep.append(a)
            return a
        b = tuple(b) + (None,) * (12 - len(b))
        texts, src_lens = np_(b[3], np.int64), np_(b[4], np.int64)
        spk_emb = None
        sa = b[2]
        if isinstance(sa, (tuple, list)) and len(sa) == 2 and isinstance(sa[1], (tuple, list)):
            # speaker_emb: dvec — batch[2] is (ref_mels, ref_slices) (collate.py:29-43); the d-vector encoder turns it into (B, d_model)
            if self.speaker_encoder is None:
                raise MttsError("batch carries (ref_mels, ref_slices) speaker args but no speaker encoder is attached (Engine.speaker_encoder)")
            sa = self.speaker_encoder(sa)
        if hasattr(sa, "detach"):
            sa = sa.detach().cpu().numpy()
        if np.ndim(sa) == 2 and np.asarray(sa).dtype.kind == "f":   # already embedded: (B, d_model) floats
            spk_emb = np_(sa, np.float32)
            assert spk_emb.shape == (texts.shape[0], self.dims.d_model), ("speaker embeddings", spk_emb.shape)
            spk = np.zeros(texts.shape[0], np.int64)
            keep.append(spk)
        else:
            spk = np_(sa, np.int64)
        if self.shared_speaker:
            spk = np.zeros_like(spk)
            keep.append(spk)
        cb = _lib.Batch()
        cb.B, cb.S_max = int(texts.shape[0]), int(b[5])
        assert texts.shape == (cb.B, cb.S_max), texts.shape
        fields = [("speakers", spk), ("texts", texts), ("src_lens", src_lens)]
        if b[11] is not None and b[6] is not None:
            mels, mel_lens = np_(b[6], np.float32), np_(b[7], np.int64)
            p, e, d = np_(b[9], np.float32), np_(b[10], np.float32), np_(b[11], np.int64)
            cb.T_max = int(b[8])
            assert mels.shape == (cb.B, cb.T_max, self.dims.n_mel), mels.shape
            assert p.shape == (cb.B, cb.T_max if self.dims.pitch_frame_level else cb.S_max), ("pitch targets", p.shape)
            assert e.shape == (cb.B, cb.T_max if self.dims.energy_frame_level else cb.S_max), ("energy targets", e.shape)
            fields += [("mels", mels), ("mel_lens", mel_lens), ("pitches", p), ("energies", e), ("durations", d)]
        if spk_emb is not None:
            fields.append(("spk_emb", spk_emb))
        for f, a in fields:
            setattr(cb, f, a.ctypes.data_as(C.c_void_p))
        return cb

    def set_batches(self, slot: int, batches: Sequence[tuple], spk_from: Optional[Sequence[tuple]] = None,
                    average_spk: bool = False):
        keep: List = []
        n = len(batches)
        if self._has_speaker_embeddings(batches[0]) or (spk_from is not None and self._has_speaker_embeddings(spk_from[0])):
            # embedded speaker args (speaker_emb: dvec / encoder): `spk_from` + `average_spk` (forward_learner's average_spk_emb,
            # base_adaptor.py:64-67) are resolved here — embed the source batch, average, expand to this batch's size
            new = []
            for i, b in enumerate(batches):
                src = spk_from[i] if spk_from is not None else b
                emb = self._embed_speaker_args(src[2])
                B = int(np.shape(b[3])[0])
                if average_spk:
                    emb = np.repeat(emb.mean(axis=0, keepdims=True), B, axis=0)
                if emb.shape[0] != B:
                    raise MttsError("speaker embedding count != batch size")
                new.append(tuple(b[:2]) + (np.ascontiguousarray(emb, np.float32),) + tuple(b[3:]))
            batches, spk_from, average_spk = new, None, False
        arr = (_lib.Batch * n)(*[self._cbatch(b, keep) for b in batches])
        sarr = None
        if spk_from is not None:
            sarr = (_lib.Batch * n)(*[self._cbatch(b, keep) for b in spk_from])
        self._ck(self.lib.mtts_set_batches(self.h, slot, n, arr, sarr, int(average_spk)))
        self.epoch += 1
        self.n_tasks[slot] = n
        self.batch_shapes[slot] = [(int(np.shape(b[3])[0]), int(b[5])) for b in batches]

    # ---- compute -----------------------------------------------------------------
    def forward(self, slot: int = 0, use_fast: bool = False, train: bool = False):
        self.epoch += 1
        self._ck(self.lib.mtts_forward(self.h, slot, int(use_fast), int(train)))

    def synthesize(self, slot: int = 0, use_fast: bool = False, train: bool = False, p_control: float = 1.0,
                   e_control: float = 1.0, d_control: float = 1.0):
        self.epoch += 1
        self._ck(self.lib.mtts_synthesize(self.h, slot, int(use_fast), int(train), p_control, e_control, d_control))

    def durations(self, slot: int = 0, task: int = 0):
        B, S = self.batch_shapes[slot][task]
        d = np.empty((B, S), np.float32)
        ml = np.empty((B,), np.int64)
        tc = C.c_int()
        self._ck(self.lib.mtts_get_durations(self.h, slot, task, d.ctypes.data_as(C.c_void_p), ml.ctypes.data_as(C.c_void_p), C.byref(tc)))
        return d, ml, int(tc.value)

    def outputs(self, slot: int = 0, task: int = 0) -> Dict[str, np.ndarray]:
        B, S = self.batch_shapes[slot][task]
        d_rounded, mel_lens, T = self.durations(slot, task)
        nm = self.dims.n_mel
        o = {"mel": np.empty((B, T, nm), np.float32), "mel_post": np.empty((B, T, nm), np.float32),
             "p": np.empty((B, T if self.dims.pitch_frame_level else S), np.float32),
             "e": np.empty((B, T if self.dims.energy_frame_level else S), np.float32), "logd": np.empty((B, S), np.float32)}
        self._ck(self.lib.mtts_get_outputs(self.h, slot, task, *[o[k].ctypes.data_as(C.c_void_p) for k in ("mel", "mel_post", "p", "e", "logd")]))
        o["d_rounded"], o["mel_lens"] = d_rounded, mel_lens
        return o

    def mel_device(self, slot: int = 0, task: int = 0, postnet: bool = True):
        """(device pointer, T_cap, floats between utterances) of the last forward's mel / mel_post of one task — lets the
        vocoder consume the synthesis in HBM (mtts_vocoder_infer_device) without a host round trip."""
        ptr, t_cap, stride = C.c_void_p(), C.c_int(), C.c_int64()
        self._ck(self.lib.mtts_get_mel_device(self.h, slot, task, int(postnet), C.byref(ptr), C.byref(t_cap), C.byref(stride)))
        return int(ptr.value), int(t_cap.value), int(stride.value)

    def adapt(self, steps: int, inner_lr: float, reset: bool = True, fetch_losses: bool = True):
        self.epoch += 1
        if fetch_losses:
            s = np.empty((steps, self.n_tasks[0], 6), np.float32)
            self._ck(self.lib.mtts_adapt(self.h, steps, inner_lr, int(reset), s.ctypes.data_as(C.c_void_p)))
            return s
        self._ck(self.lib.mtts_adapt(self.h, steps, inner_lr, int(reset), None))
        return None

    def loss(self, slot: int = 0) -> np.ndarray:
        out = np.empty((self.n_tasks[slot], 6), np.float32)
        self._ck(self.lib.mtts_loss(self.h, slot, out.ctypes.data_as(C.c_void_p)))
        return out

    def backward(self, slot: int = 0, use_fast: bool = False, scale: float = 1.0, need_encoder: bool = True):
        self._ck(self.lib.mtts_backward(self.h, slot, int(use_fast), float(scale), int(need_encoder)))

    def meta_grad(self, steps: int, inner_lr: float, grad_scale: float, second_order: bool = False, fetch_losses: bool = True):
        self.epoch += 1
        nt = self.n_tasks[0]
        if fetch_losses:
            q = np.empty((nt, 6), np.float32)
            s = np.empty((steps, nt, 6), np.float32)
            self._ck(self.lib.mtts_meta_grad(self.h, steps, inner_lr, grad_scale, int(second_order),
                                             q.ctypes.data_as(C.c_void_p), s.ctypes.data_as(C.c_void_p)))
            return q, s
        self._ck(self.lib.mtts_meta_grad(self.h, steps, inner_lr, grad_scale, int(second_order), None, None))
        return None, None

    def hvp_support(self):
        """H v of the support loss at the current fast weights, v = per-task gradient buffer (export with which=6)."""
        self._ck(self.lib.mtts_hvp_support(self.h))

    # ---- iMAML (include/mtts.h) ---------------------------------------------------
    def set_inner_prox(self, reg_param: float):
        self._ck(self.lib.mtts_set_inner_prox(self.h, float(reg_param)))

    def imaml_begin(self) -> np.ndarray:
        self.epoch += 1
        q = np.empty((self.n_tasks[1], 6), np.float32)
        self._ck(self.lib.mtts_imaml_begin(self.h, q.ctypes.data_as(C.c_void_p)))
        return q

    def imaml_cg_step(self, inner_lr: float, reg_param: float, tol: float = 1e-10):
        self.epoch += 1
        self._ck(self.lib.mtts_imaml_cg_step(self.h, inner_lr, reg_param, tol))

    def imaml_finish(self, inner_lr: float, reg_param: float, grad_scale: float, max_norm: float = 0.0) -> np.ndarray:
        norms = np.empty((self.n_tasks[0],), np.float32)
        self._ck(self.lib.mtts_imaml_finish(self.h, inner_lr, reg_param, grad_scale, max_norm, norms.ctypes.data_as(C.c_void_p)))
        return norms

    def plain_grad(self, slot: int = 0, grad_scale: float = 1.0, fetch_losses: bool = True):
        self.epoch += 1
        if fetch_losses:
            q = np.empty((self.n_tasks[slot], 6), np.float32)
            self._ck(self.lib.mtts_plain_grad(self.h, slot, grad_scale, q.ctypes.data_as(C.c_void_p)))
            return q
        self._ck(self.lib.mtts_plain_grad(self.h, slot, grad_scale, None))
        return None

    def outer_grad_ptr(self) -> int:
        return int(self.lib.mtts_outer_grad_ptr(self.h))

    def outer_grad_view(self):
        """Device buffer a host-side collective reduces, as an object torch.as_tensor(..., device='cuda') can alias: the outer gradient
        followed by the exchange tail (loss scalars + BatchNorm running buffers; bracket the collective with sync_pack / sync_unpack)."""
        return _DevView(self.outer_grad_ptr(), self.sync_floats)

    @property
    def sync_floats(self) -> int:
        return int(self.lib.mtts_outer_sync_floats(self.h))

    def sync_pack(self, bn_weight: float = 1.0):
        """Fill the exchange tail behind the outer gradient (include/mtts.h): this rank's scaled loss sums and its BatchNorm running
        buffers x bn_weight (rank 0: 1, others: 0 = DDP's broadcast_buffers; 1 / world = mean)."""
        self._ck(self.lib.mtts_sync_pack(self.h, float(bn_weight)))

    def sync_unpack(self):
        self._ck(self.lib.mtts_sync_unpack(self.h))

    def synced_losses(self) -> np.ndarray:
        """The six loss scalars reduced over the ranks by the last exchange (log_dict(sync_dist=True), meta.py:78-79)."""
        out = np.empty(6, np.float32)
        self._ck(self.lib.mtts_get_synced_losses(self.h, out.ctypes.data_as(C.c_void_p)))
        return out

    def set_bn_sync(self, mode: str = "rank0"):
        self._ck(self.lib.mtts_set_bn_sync(self.h, {"rank0": 0, "mean": 1}[mode]))
        self.bn_sync = mode

    def bn_pack_weight(self, rank: int, world: int) -> float:
        """Weight of this rank's BatchNorm running buffers in the exchange tail when the collective is issued OUTSIDE the library
        (sync_pack / all_reduce / sync_unpack): the same rule mtts_allreduce_outer applies — "rank0": rank 0's buffers (DDP
        broadcast_buffers, main.py:32), "mean": 1 / world."""
        if getattr(self, "bn_sync", "rank0") == "mean":
            return 1.0 / float(world)
        return 1.0 if rank == 0 else 0.0

    # ---- RCCL inside the library (include/mtts.h: mtts_comm_*) -------------------
    def comm_available(self) -> bool:
        """True when librccl can be loaded by this process (a local probe, no communication)."""
        return self.lib.mtts_comm_available(self.h) == 0

    def comm_unique_id(self) -> bytes:
        buf = C.create_string_buffer(128)
        self._ck(self.lib.mtts_comm_unique_id(self.h, buf))
        return buf.raw

    def comm_init(self, unique_id: bytes, rank: int, world_size: int):
        assert len(unique_id) == 128
        self._ck(self.lib.mtts_comm_init(self.h, C.create_string_buffer(unique_id, 128), rank, world_size))

    def arm_allreduce_overlap(self) -> bool:
        """Arm the overlapped, bucketed exchange for the NEXT gradient call (include/mtts.h); False: not available, allreduce_outer will
        reduce the whole buffer in one collective as before."""
        rc = self.lib.mtts_arm_allreduce_overlap(self.h)
        if rc < 0:
            self._ck(rc)
        return rc == 0

    def disarm_allreduce_overlap(self):
        """Take the arming back (the gradient call it was meant for will not happen); the gradient calls disarm on every exit path themselves."""
        self._ck(self.lib.mtts_disarm_allreduce_overlap(self.h))

    @property
    def allreduce_bucket_agreement(self) -> int:
        """1: every rank holds the same bucket table (agreed collectively in comm_init); 0: they disagreed, the overlap is off on all ranks;
        -1: no communicator."""
        return int(self.lib.mtts_allreduce_bucket_agreement(self.h))

    @property
    def allreduce_launches(self) -> int:
        return int(self.lib.mtts_allreduce_launches(self.h))

    @property
    def inner_update_launches(self) -> int:
        """Launches of the last inner SGD step: > 1 when it ran module by module behind its backward, 0 for the single pass."""
        return int(self.lib.mtts_inner_update_launches(self.h))

    def allreduce_outer(self):
        """ncclAllReduce(SUM) of the outer-gradient buffer on the engine's stream (asynchronous)."""
        self._ck(self.lib.mtts_allreduce_outer(self.h))

    def outer_update(self, lr: float, betas=(0.9, 0.98), eps: float = 1e-9, weight_decay: float = 0.0,
                     max_norm: float = 1.0, grad_ptr: Optional[int] = None, fetch_norm: bool = False):
        norm = C.c_float()
        self._ck(self.lib.mtts_outer_update(self.h, C.c_void_p(grad_ptr) if grad_ptr else None, lr, betas[0], betas[1], eps,
                                            weight_decay, max_norm, C.byref(norm) if fetch_norm else None))
        return float(norm.value) if fetch_norm else None

    # hooks for a module trained in front of the engine (the LSTM speaker encoder)
    def speaker_grad(self, task: int, B: int) -> np.ndarray:
        out = np.empty((B, self.dims.d_model), np.float32)
        self._ck(self.lib.mtts_get_speaker_grad(self.h, task, B, out.ctypes.data_as(C.c_void_p)))
        return out

    def set_extra_grad_sumsq(self, dev_ptr):
        self._ck(self.lib.mtts_set_extra_grad_sumsq(self.h, C.c_void_p(dev_ptr) if dev_ptr else None))

    def grad_norm_ptr(self):
        return C.c_void_p(self.lib.mtts_grad_norm_dev(self.h))

    def reset_optimizer(self):
        self._ck(self.lib.mtts_reset_optimizer(self.h))
