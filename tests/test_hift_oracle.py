"""The HiFT vocoder oracle (oracle/hift_oracle.py) against tests/golden/hift_fixture.npz, which holds outputs of the
reference's own HiFTGenerator / ConvRNNF0Predictor classes on the package's seeded weights
(tests/golden/make_hift_fixtures.py).  CPU only."""
import ctypes as C
import math
import os

import numpy as np
import pytest
import torch

from chatterbox_vllm_amd import hift_weights as HW
from oracle import hift_oracle as HO

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def fx():
    return np.load(os.path.join(ROOT, "tests", "golden", "hift_fixture.npz"))


@pytest.fixture(scope="module")
def w():
    return HO.fold_weight_norm(HW.synthetic_state_dict())


def reference_draws(seed, B, T):
    """The two random draws of SineGen.forward in the order the reference makes them (hifigan.py:212-213, 226) from
    torch's global generator."""
    torch.manual_seed(seed)
    phase = torch.distributions.uniform.Uniform(low=-np.pi, high=np.pi).sample(sample_shape=(B, 9, 1))
    noise = torch.randn(B, 9, T * HW.SAMPLES_PER_FRAME)
    return phase, noise


def handmade_source(n_frames, batch, seed):
    g = torch.Generator().manual_seed(seed)
    L = n_frames * HW.SAMPLES_PER_FRAME
    t = torch.arange(L, dtype=torch.float32) / HW.SR
    s = 0.6 * torch.sin(2 * torch.pi * (110.0 + 400.0 * t) * t) + 0.05 * torch.randn(batch, L, generator=g)
    return s.view(batch, 1, L).contiguous()


def test_inventory_and_fold_match_the_package(w):
    folded = HW.fold_weight_norm(HW.synthetic_state_dict())
    assert list(folded) == list(w)
    for k in w:
        assert torch.equal(folded[k], w[k]), k
    n = sum(math.prod(s) for _, s in HW.tensor_inventory())
    assert n == 20821295                       # parameter count of the reference's mel2wav state dict


def test_f0_matches_the_reference(fx, w):
    mel = HW.synthetic_mel(int(fx["a_frames"]), seed=int(fx["a_mel_seed"]))
    f0 = HO.f0_predict(w, mel)
    ref = torch.from_numpy(fx["a_f0"])
    assert (f0 - ref).abs().max().item() <= 2e-4 * ref.abs().max().item()
    assert (ref > HO.VOICED_THRESHOLD).any() and ref.min() >= 0


def test_source_matches_the_reference_with_its_own_draws(fx, w):
    T = int(fx["a_frames"])
    phase, noise = reference_draws(int(fx["a_rng_seed"]), 1, T)
    s = HO.harmonic_source(w, torch.from_numpy(fx["a_f0"]), phase, noise)
    ref = torch.from_numpy(fx["a_source"])
    assert s.shape == ref.shape
    assert (s - ref).abs().max().item() <= 1e-5


def test_stft_is_torch_stft(fx):
    s2 = handmade_source(int(fx["b_frames"]), 2, int(fx["b_source_seed"]))
    re, im = HO.stft(s2.squeeze(1), torch.float32)
    assert (re - torch.from_numpy(fx["b_stft_re"])).abs().max().item() <= 2e-6 * 5.1
    assert (im - torch.from_numpy(fx["b_stft_im"])).abs().max().item() <= 2e-6 * 5.1


def test_decode_matches_the_reference(fx, w):
    T2 = int(fx["b_frames"])
    mel2 = HW.synthetic_mel(T2, seed=int(fx["b_mel_seed"]), batch=2)
    s2 = handmade_source(T2, 2, int(fx["b_source_seed"]))
    taps = {}
    wav = HO.decode(w, mel2, s2, taps=taps)
    ref = torch.from_numpy(fx["b_wav"])
    assert (taps["conv_pre"] - torch.from_numpy(fx["b_conv_pre"])).abs().max().item() <= 1e-4
    assert (taps["conv_post"] - torch.from_numpy(fx["b_conv_post"])).abs().max().item() <= 1e-4
    assert wav.shape == ref.shape
    assert (wav - ref).abs().max().item() <= 2e-5
    # fp64 restatement: the reference's fp32 result sits within rounding of it
    w64 = {k: v.double() for k, v in w.items()}
    wav64 = HO.decode(w64, mel2, s2, dtype=torch.float64)
    assert (wav64.float() - ref).abs().max().item() <= 5e-5


def test_inference_matches_the_reference(fx, w):
    T = int(fx["a_frames"])
    mel = HW.synthetic_mel(T, seed=int(fx["a_mel_seed"]))
    phase, noise = reference_draws(int(fx["a_rng_seed"]), 1, T)
    wav, s = HO.inference(w, mel, phase, noise)
    # 5e-5 on a waveform of amplitude 0.08: the source already differs by up to 1e-5 (the running phase sum is taken in
    # another order), and the network passes that on
    assert (wav - torch.from_numpy(fx["a_wav"])).abs().max().item() <= 5e-5
    cache = torch.from_numpy(fx["a_cache"])
    wav_c, s_c = HO.inference(w, mel, phase, noise, cache_source=cache)
    assert torch.equal(s_c[:, :, :600], cache)
    assert (wav_c - torch.from_numpy(fx["a_wav_cached"])).abs().max().item() <= 5e-5


def test_library_loads_and_exports_every_declared_symbol():
    """libs3hift.so exports every function include/s3hift_hip.h, include/s3flow_hip.h and include/s3cfm_hip.h declare, and
    the bindings declare the same sets."""
    import re
    from chatterbox_vllm_amd import cfm_decoder, flow_encoder, hift
    lib = C.CDLL(hift.lib_path())
    for header, prefix, sigs in (("s3hift_hip.h", "s3hift_", hift._SIGNATURES), ("s3flow_hip.h", "s3flow_", flow_encoder._SIGNATURES),
                                 ("s3cfm_hip.h", "s3cfm_", cfm_decoder._SIGNATURES)):
        text = open(os.path.join(ROOT, "include", header)).read()
        declared = set(re.findall(r"\b(" + prefix + r"[a-z0-9_]+)\s*\(", text))
        assert declared == set(sigs), declared ^ set(sigs)
        for name in declared:
            assert hasattr(lib, name), name
    assert lib.s3hift_abi_version() == hift.ABI_VERSION
