"""ctypes binding of libmtts.so (include/mtts.h).

The product path loads exactly one thing: the HIP library built for gfx950 that sits next to this
file.  If it is missing this module raises — there is no CPU fallback.  (The test-suite's SIMT
emulator build of the same sources, tests/emu/libmtts_emu.so, can only be reached by passing its
path explicitly to :func:`load`, which nothing in this package does.)
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(HERE, "libmtts.so")


class ModelCfg(C.Structure):
    _fields_ = [(n, C.c_int) for n in (
        "d_model", "enc_layers", "dec_layers", "enc_heads", "dec_heads", "d_ff", "k1", "k2",
        "vp_filter", "vp_kernel", "n_bins", "max_seq_len", "n_mel", "vocab", "n_speaker",
        "postnet_dim", "postnet_kernel", "postnet_layers")] + [
        ("pitch_min", C.c_float), ("pitch_max", C.c_float), ("energy_min", C.c_float), ("energy_max", C.c_float),
        ("adapt_mask", C.c_int), ("enc_dropout", C.c_float), ("dec_dropout", C.c_float), ("vp_dropout", C.c_float),
        ("pitch_frame_level", C.c_int), ("energy_frame_level", C.c_int)]


class Batch(C.Structure):
    _fields_ = [("B", C.c_int), ("S_max", C.c_int), ("T_max", C.c_int),
                ("speakers", C.c_void_p), ("texts", C.c_void_p), ("src_lens", C.c_void_p),
                ("mels", C.c_void_p), ("mel_lens", C.c_void_p), ("pitches", C.c_void_p),
                ("energies", C.c_void_p), ("durations", C.c_void_p), ("spk_emb", C.c_void_p)]


EXPORTS = {
    # name: (restype, argtypes)
    "mtts_create": (C.c_int, [C.POINTER(ModelCfg), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "mtts_destroy": (None, [C.c_void_p]),
    "mtts_last_error": (C.c_char_p, [C.c_void_p]),
    "mtts_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mtts_set_dropout": (C.c_int, [C.c_void_p, C.c_int, C.c_uint]),
    "mtts_set_numerics": (C.c_int, [C.c_void_p, C.c_int]),
    "mtts_get_numerics": (C.c_int, [C.c_void_p]),
    "mtts_synchronize": (C.c_int, [C.c_void_p]),
    "mtts_param_count": (C.c_int, [C.c_void_p]),
    "mtts_param_info": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_int), C.POINTER(C.c_int * 4),
                                  C.POINTER(C.c_int64), C.POINTER(C.c_int)]),
    "mtts_param_total": (C.c_int64, [C.c_void_p]),
    "mtts_adapt_start": (C.c_int64, [C.c_void_p]),
    "mtts_load_param": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int64]),
    "mtts_import_state": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int, C.c_void_p, C.c_int64]),
    "mtts_set_optimizer_step": (C.c_int, [C.c_void_p, C.c_int64]),
    "mtts_export_param": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.c_void_p, C.c_int64]),
    "mtts_set_bn_buffers": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int64]),
    "mtts_get_bn_buffers": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64)]),
    "mtts_set_bins": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "mtts_get_bins": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "mtts_set_batches": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(Batch), C.POINTER(Batch), C.c_int]),
    "mtts_forward": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "mtts_synthesize": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float]),
    "mtts_get_durations": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]),
    "mtts_adapt": (C.c_int, [C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_void_p]),
    "mtts_get_outputs": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mtts_loss": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "mtts_backward": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_int]),
    "mtts_meta_grad": (C.c_int, [C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_int, C.c_void_p, C.c_void_p]),
    "mtts_hvp_support": (C.c_int, [C.c_void_p]),
    "mtts_reserve_second_order": (C.c_int, [C.c_void_p, C.c_int]),
    "mtts_set_inner_prox": (C.c_int, [C.c_void_p, C.c_float]),
    "mtts_imaml_begin": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mtts_imaml_cg_step": (C.c_int, [C.c_void_p, C.c_float, C.c_float, C.c_float]),
    "mtts_imaml_finish": (C.c_int, [C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p]),
    "mtts_plain_grad": (C.c_int, [C.c_void_p, C.c_int, C.c_float, C.c_void_p]),
    "mtts_outer_grad_ptr": (C.c_void_p, [C.c_void_p]),
    "mtts_comm_available": (C.c_int, [C.c_void_p]),
    "mtts_comm_unique_id": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mtts_comm_init": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int]),
    "mtts_allreduce_outer": (C.c_int, [C.c_void_p]),
    "mtts_arm_allreduce_overlap": (C.c_int, [C.c_void_p]),
    "mtts_disarm_allreduce_overlap": (C.c_int, [C.c_void_p]),
    "mtts_allreduce_bucket_agreement": (C.c_int, [C.c_void_p]),
    "mtts_allreduce_launches": (C.c_int, [C.c_void_p]),
    "mtts_inner_update_launches": (C.c_int, [C.c_void_p]),
    "mtts_outer_sync_floats": (C.c_int64, [C.c_void_p]),
    "mtts_sync_pack": (C.c_int, [C.c_void_p, C.c_float]),
    "mtts_sync_unpack": (C.c_int, [C.c_void_p]),
    "mtts_get_synced_losses": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mtts_set_bn_sync": (C.c_int, [C.c_void_p, C.c_int]),
    "mtts_outer_update": (C.c_int, [C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float,
                                    C.c_float, C.c_void_p]),
    "mtts_reset_optimizer": (C.c_int, [C.c_void_p]),
    "mtts_set_grad_accumulation": (C.c_int, [C.c_void_p, C.c_int]),
    "mtts_profile_gemm": (C.c_int, [C.c_void_p, C.c_int]),
    "mtts_profile_kinds": (C.c_int, []),
    "mtts_profile_kernel_name": (C.c_char_p, [C.c_int]),
    "mtts_profile_report": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.c_int]),
    "mtts_gemm_f32": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                                C.c_int, C.c_void_p, C.c_float, C.c_int, C.c_int, C.c_void_p]),
    "mtts_xcd_schedule_check": (C.c_int, [C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)]),
    "mtts_gemm_f32_dual": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_int, C.c_void_p, C.c_float, C.c_int, C.c_int, C.c_void_p]),
    "mtts_to_bf16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_longlong, C.c_void_p]),
    "mtts_gemm_bf16_planes": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                        C.c_void_p, C.c_float, C.c_int, C.c_int, C.c_void_p]),
    "mtts_plane_problems": (C.c_longlong, [C.c_void_p]),
    "mtts_gemm_bf16": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                 C.c_void_p, C.c_int, C.c_void_p, C.c_float, C.c_int, C.c_int, C.c_void_p]),
    "mtts_conv1d_f32": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_int, C.c_void_p]),
    "mtts_kernel_ws_bytes": (C.c_int64, [C.c_int, C.c_int]),
    "mtts_layernorm_fwd": (C.c_int, [C.c_int, C.c_int] + [C.c_void_p] * 10),
    "mtts_layernorm_bwd": (C.c_int, [C.c_int, C.c_int] + [C.c_void_p] * 10),
    "mtts_softmax_fwd": (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mtts_softmax_bwd": (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p]),
    "mtts_sdpa_fwd": (C.c_int, [C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 7),
    "mtts_batchnorm_fwd": (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_void_p]),
    "mtts_batchnorm_bwd": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mtts_table_grad": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mtts_length_regulate_fwd": (C.c_int, [C.c_int, C.c_int] + [C.c_void_p] * 8),
    "mtts_length_regulate_bwd": (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "mtts_layernorm_jvp": (C.c_int, [C.c_int, C.c_int] + [C.c_void_p] * 11),
    "mtts_softmax_jvp": (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mtts_dvector_create": (C.c_int, [C.c_int] * 8 + [C.POINTER(C.c_void_p)]),
    "mtts_dvector_destroy": (None, [C.c_void_p]),
    "mtts_dvector_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mtts_dvector_last_error": (C.c_char_p, [C.c_void_p]),
    "mtts_dvector_load": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int64]),
    "mtts_dvector_embed": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "mtts_dvector_enable_training": (C.c_int, [C.c_void_p]),
    "mtts_dvector_embed_train": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "mtts_dvector_backward": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mtts_dvector_grad_sumsq": (C.c_void_p, [C.c_void_p]),
    "mtts_dvector_adam_step": (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_float] * 6),
    "mtts_dvector_set_optimizer_step": (C.c_int, [C.c_void_p, C.c_int]),
    "mtts_dvector_export": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int, C.c_void_p, C.c_int64]),
    "mtts_dvector_import": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int, C.c_void_p, C.c_int64]),
    "mtts_get_speaker_grad": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "mtts_set_extra_grad_sumsq": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mtts_grad_norm_dev": (C.c_void_p, [C.c_void_p]),
    "mtts_stft_create": (C.c_int, [C.c_int] * 5 + [C.POINTER(C.c_void_p)]),
    "mtts_stft_destroy": (None, [C.c_void_p]),
    "mtts_stft_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mtts_stft_last_error": (C.c_char_p, [C.c_void_p]),
    "mtts_stft_load": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "mtts_stft_mel_spectrogram": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "mtts_vocoder_create": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int, C.c_int,
                                      C.POINTER(C.c_void_p)]),
    "mtts_vocoder_destroy": (None, [C.c_void_p]),
    "mtts_vocoder_last_error": (C.c_char_p, [C.c_void_p]),
    "mtts_vocoder_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mtts_vocoder_hop": (C.c_int, [C.c_void_p]),
    "mtts_vocoder_param_count": (C.c_int, [C.c_void_p]),
    "mtts_vocoder_param_info": (C.c_int, [C.c_void_p, C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_int64)]),
    "mtts_vocoder_load": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int64]),
    "mtts_vocoder_infer": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_float, C.c_void_p]),
    "mtts_vocoder_infer_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_float, C.c_void_p]),
    "mtts_get_mel_device": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.POINTER(C.c_int64)]),
}

_cache = {}


def load(path: str | None = None) -> C.CDLL:
    """Load libmtts and type every entry point of include/mtts.h.  Raises if the library (or a
    symbol) is missing."""
    path = os.path.abspath(path or DEFAULT_LIB)
    if path in _cache:
        return _cache[path]
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} not found: build the HIP library first (python -c 'import __graft_entry__ as g; g.build()'). "
            "meta_tts_amd has no CPU fallback.")
    # libmtts links the ROCm HIP runtime by soname.  torch ships its own copy of that runtime; if
    # torch is going to be used in this process (device memory, streams, torch.distributed) it must
    # be imported first so both share ONE runtime instance (two instances cannot both own the GPU).
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(path)
    for name, (res, args) in EXPORTS.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _cache[path] = lib
    return lib
