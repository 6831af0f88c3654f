"""MelGAN generator: host packing logic + the HIP orchestration through the SIMT emulator on
CPU, and the real kernels on the GPU, against oracle/melgan_oracle.py (parity unpinned: the generator is an un-vendored
torch.hub dependency of the reference — see the oracle header)."""
import math
import os

import numpy as np
import pytest

import __graft_entry__ as ge
from meta_tts_amd import vocoder as V
from oracle import melgan_oracle as MO


@pytest.fixture(scope="module")
def emu_lib():
    return ge.build_emulator()


def _mel(seed, B, T, n_mel):
    g = np.random.RandomState(seed)
    return (g.standard_normal((B, n_mel, T)) * 1.5 - 4.0).astype(np.float32)


def test_spec_matches_hub_module_numbering():
    names = [n for n, _, _ in V.generator_spec()]
    # ReflectionPad(0) Conv(1) | LeakyReLU(2) ConvT(3) Res(4,5,6) | (7) ConvT(8) Res(9,10,11) | (12) ConvT(13) Res(14,15,16)
    # | (17) ConvT(18) Res(19,20,21) | LeakyReLU(22) ReflectionPad(23) Conv(24) Tanh(25)
    assert names[0] == "model.1" and names[1] == "model.3" and names[2] == "model.4.block.2"
    assert names[-1] == "model.24" and "model.18" in names and "model.21.shortcut" in names
    sd = V.synthetic_state_dict(0)
    assert sd["model.3.weight_v"].shape == (512, 256, 16) and sd["model.24.weight_v"].shape == (1, 32, 7)
    assert sum(v.size for k, v in sd.items() if k.endswith("weight_v")) > 4_000_000   # ~4.3 M parameters


def test_polyphase_image_equals_conv_transpose():
    import torch
    import torch.nn.functional as F
    g = np.random.RandomState(3)
    cin, cout, r, T = 8, 4, 4, 11
    w = g.standard_normal((cin, cout, 2 * r)).astype(np.float32)
    x = g.standard_normal((T, cin)).astype(np.float32)
    ref = F.conv_transpose1d(torch.from_numpy(x.T[None]), torch.from_numpy(w), stride=r, padding=r // 2).numpy()[0].T  # [rT][cout]
    p = r // 2
    xz = np.concatenate([np.zeros((1, cin), np.float32), x, np.zeros((1, cin), np.float32)])
    out = np.zeros((r * T, cout), np.float32)
    for ph in range(r):
        img = np.concatenate([w[:, :, ph + r].T, w[:, :, ph].T], axis=1)      # [cout][2 cin], as pack_tensors builds it
        q0 = 0 if ph >= p else 1
        for m in range(T):
            q = q0 + m
            out[q * r + ph - p] = img @ np.concatenate([xz[q], xz[q + 1]])
    np.testing.assert_allclose(out, ref, rtol=1e-5, atol=1e-5)


def test_tiny_generator_through_emulator(emu_lib):
    kw = dict(n_mel=16, ngf=16, n_res=2, ratios=(4, 2))
    sd = V.synthetic_state_dict(5, **kw)
    voc = V.MelGAN(sd, max_B=2, max_T=24, lib_path=emu_lib, **kw)
    mel = _mel(1, 2, 20, 16)
    lens = np.array([20, 13], np.int32)
    wav = voc.mel2wav(mel, lens, mel_scale=1.0 / math.log(10.0))
    assert wav.shape == (2, 20 * 8)
    ref0 = MO.mel2wav(sd, mel[:1] / math.log(10.0), n_res=2, ratios=(4, 2))[0]
    ref1 = MO.mel2wav(sd, mel[1:, :, :13] / math.log(10.0), n_res=2, ratios=(4, 2))[0]
    np.testing.assert_allclose(wav[0], ref0, atol=2e-5)
    np.testing.assert_allclose(wav[1, :13 * 8], ref1, atol=2e-5)   # ragged: reflection pads at the utterance's own end
    assert np.all(wav[1, 13 * 8:] == 0)
    i16 = voc.infer(mel, 32768.0, lengths=[20 * 8, 13 * 8 - 3])
    ref_i16 = MO.infer(sd, mel[:1], 32768.0, n_res=2, ratios=(4, 2))[0]
    assert i16[0].dtype == np.int16 and len(i16[1]) == 13 * 8 - 3
    assert np.abs(i16[0].astype(np.int32) - ref_i16.astype(np.int32)).max() <= 2
    voc.close()


@pytest.mark.gpu
def test_full_generator_on_gpu_vs_oracle():
    sd = V.synthetic_state_dict(0)
    voc = V.MelGAN(sd, max_B=2, max_T=96)
    mel = _mel(2, 2, 80, 80)
    lens = np.array([80, 57], np.int32)
    wav = voc.mel2wav(mel, lens, mel_scale=1.0 / math.log(10.0))
    ref0 = MO.mel2wav(sd, mel[:1] / math.log(10.0))[0]
    ref1 = MO.mel2wav(sd, mel[1:, :, :57] / math.log(10.0))[0]
    assert wav.shape == (2, 80 * 256) and np.abs(ref0).max() > 0.05 and np.abs(ref0).max() < 1.0
    np.testing.assert_allclose(wav[0], ref0, atol=1e-4)
    np.testing.assert_allclose(wav[1, :57 * 256], ref1, atol=1e-4)
    wav2 = voc.mel2wav(mel, lens, mel_scale=1.0 / math.log(10.0))
    assert np.array_equal(wav, wav2)   # run-to-run identical
    voc.close()


@pytest.mark.gpu
def test_device_pointer_entry_matches_host_entry():
    """mtts_vocoder_infer_device: mel and waveform stay in HBM (torch tensors only provide the memory)."""
    import ctypes as C
    import torch
    sd = V.synthetic_state_dict(0)
    voc = V.MelGAN(sd, max_B=2, max_T=64)
    mel = _mel(4, 2, 48, 80)
    lens = np.array([48, 31], np.int32)
    ref = voc.mel2wav(mel, lens, mel_scale=0.5)
    voc.set_stream(torch.cuda.current_stream().cuda_stream)
    x = torch.from_numpy(np.ascontiguousarray(mel.transpose(0, 2, 1))).cuda()      # [B][T][n_mel]
    wav = torch.zeros(2, 48 * voc.hop, device="cuda")
    rc = voc.lib.mtts_vocoder_infer_device(voc.h, C.c_void_p(x.data_ptr()), 0, 2, 48, lens.ctypes.data_as(C.c_void_p), C.c_float(0.5),
                                           C.c_void_p(wav.data_ptr()))
    assert rc == 0
    torch.cuda.synchronize()
    got = wav.cpu().numpy()
    np.testing.assert_array_equal(got[0], ref[0])
    np.testing.assert_array_equal(got[1, :31 * voc.hop], ref[1, :31 * voc.hop])
    voc.close()


@pytest.mark.gpu
def test_engine_mel_feeds_the_vocoder_in_hbm():
    """mtts_get_mel_device -> mtts_vocoder_infer_device: the synthesised mel_post never leaves the GPU; same waveform as
    downloading it and using the host entry."""
    import torch
    from meta_tts_amd import synth
    from meta_tts_amd.config import ModelDims
    from meta_tts_amd.engine import Engine
    dims = ModelDims()
    b = synth.make_batch(3, 2, speaker=4, s_range=(10, 20), d_range=(2, 6), first_len=16)
    eng = Engine(dims, adapt_modules=[], max_tasks=1, max_B=2, max_S=20, max_T=int(b[8]))
    eng.load_params(synth.make_params(dims, 0))
    eng.set_batches(0, [b])
    eng.forward(0, train=False)
    out = eng.outputs(0, 0)
    lens = np.asarray(b[7], np.int32)
    voc = V.MelGAN(max_B=2, max_T=int(b[8]))
    ref = voc.mel2wav(np.ascontiguousarray(out["mel_post"].transpose(0, 2, 1)), lens, mel_scale=0.4)
    ptr, tcap, stride = eng.mel_device(0, 0, postnet=True)
    assert tcap == int(b[8]) and stride > tcap * dims.n_mel
    wav = torch.zeros(2, tcap * voc.hop, device="cuda")
    voc.mel2wav_device(ptr, stride, 2, tcap, lens, wav.data_ptr(), mel_scale=0.4)
    torch.cuda.synchronize()
    got = wav.cpu().numpy()
    for i in range(2):
        np.testing.assert_array_equal(got[i, :lens[i] * voc.hop], ref[i, :lens[i] * voc.hop])
    voc.close(); eng.close()
