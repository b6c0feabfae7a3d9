"""HiFT vocoder on the GPU (lib/libs3hift.so through its C ABI) against the reference's outputs
(tests/golden/hift_fixture.npz) and the CPU oracle (oracle/hift_oracle.py, itself held to that fixture by
tests/test_hift_oracle.py).  All arithmetic is fp32 on both sides; tolerances are absolute, stated per test, on
waveforms of amplitude ~0.1 and activations of order one."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from chatterbox_vllm_amd import hift_weights as HW

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV = "cuda:0"


@pytest.fixture(scope="module")
def fx():
    return np.load(os.path.join(ROOT, "tests", "golden", "hift_fixture.npz"))


@pytest.fixture(scope="module")
def gen():
    from chatterbox_vllm_amd.hift import HiFTGenerator
    g = HiFTGenerator(HW.synthetic_state_dict(), max_batch=2, max_frames=320)
    yield g
    g.shutdown()


@pytest.fixture(scope="module")
def w():
    from oracle import hift_oracle as HO
    return HO.fold_weight_norm(HW.synthetic_state_dict())


def handmade_source(n_frames, batch, seed):
    g = torch.Generator().manual_seed(seed)
    L = n_frames * HW.SAMPLES_PER_FRAME
    t = torch.arange(L, dtype=torch.float32) / HW.SR
    s = 0.6 * torch.sin(2 * torch.pi * (110.0 + 400.0 * t) * t) + 0.05 * torch.randn(batch, L, generator=g)
    return s.view(batch, 1, L).contiguous()


def reference_draws(seed, B, T):
    torch.manual_seed(seed)
    phase = torch.distributions.uniform.Uniform(low=-np.pi, high=np.pi).sample(sample_shape=(B, 9, 1))
    noise = torch.randn(B, 9, T * HW.SAMPLES_PER_FRAME)
    return phase, noise


CONV_CASES = [
    # B, Cin, Cout, K, dil, T, act, with_res
    (1, 64, 64, 3, 1, 300, 1, True),
    (2, 256, 256, 11, 5, 129, 1, True),
    (1, 128, 128, 7, 3, 1, 1, False),        # a single position: everything but one tap is padding
    (2, 80, 512, 7, 1, 37, 0, False),        # conv_pre
    (1, 64, 18, 7, 1, 513, 2, False),        # conv_post: 18 channels in a block of 32, the 32 x 256 tile form
    (1, 512, 512, 3, 1, 128, 3, False),      # f0 condnet with ELU
    (1, 16, 96, 1, 1, 255, 0, False),        # output channels not a multiple of 64
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv1d_kernel_against_fp64(case):
    """One launch of the MFMA convolution kernel against the same convolution in fp64 on the CPU: the error is that of
    an fp32 sum of Cin x K products (1e-6 relative to the sum of magnitudes), far below 2e-5 of the output's scale."""
    from chatterbox_vllm_amd.hift import conv1d_op
    B, Cin, Cout, K, dil, T, act, with_res = case
    g = torch.Generator().manual_seed(Cin * 1000 + K * 10 + dil)
    x = torch.randn(B, Cin, T, generator=g)
    wt = torch.randn(Cout, Cin, K, generator=g) / (Cin * K) ** 0.5
    bias = torch.randn(Cout, generator=g)
    alpha = 0.5 + torch.rand(Cin, generator=g)
    pad = dil * (K - 1) // 2
    res = torch.randn(B, Cout, T, generator=g) if with_res else None
    xa = x.double()
    if act == 1:
        a = alpha.double().view(1, -1, 1)
        xa = xa + (1.0 / (a + 1e-9)) * torch.sin(xa * a) ** 2
    elif act == 2:
        xa = F.leaky_relu(xa, 0.1)
    elif act == 3:
        xa = F.elu(xa)
    want = F.conv1d(xa, wt.double(), bias.double(), dilation=dil, padding=pad)
    if with_res:
        want = want + res.double()
    got = conv1d_op(x.to(DEV), wt, bias, dilation=dil, padding=pad, act=act, alpha=alpha.to(DEV), slope=0.1,
                    res=res.to(DEV) if with_res else None).cpu()
    assert got.shape == want.shape
    err = (got.double() - want).abs().max().item()
    assert err <= 2e-5 * max(1.0, want.abs().max().item()), err


def test_f0_matches_the_reference(gen, fx):
    mel = HW.synthetic_mel(int(fx["a_frames"]), seed=int(fx["a_mel_seed"]))
    f0 = gen.f0_predictor(mel).cpu()
    ref = torch.from_numpy(fx["a_f0"])
    assert (f0 - ref).abs().max().item() <= 2e-4 * ref.abs().max().item()      # f0 of ~85 Hz: 0.017 Hz


def test_source_with_the_references_draws(gen, fx, w):
    """With the reference's own random draws handed in, the source equals the reference's to 1e-6 (observed 9e-8): the
    running phase is a closed form per frame in fp64 here, a running sum in double stored as fp32 there (torch.cumsum on
    the CPU), and both take the fractional part of the fp32 value."""
    from oracle import hift_oracle as HO
    T = int(fx["a_frames"])
    phase, noise = reference_draws(int(fx["a_rng_seed"]), 1, T)
    f0 = torch.from_numpy(fx["a_f0"])
    s = gen.source(f0, phase=phase, noise=noise).cpu()
    err = (s - torch.from_numpy(fx["a_source"])).abs().max().item()
    print("source vs reference: max |err| %.3g" % err)
    assert err <= 1e-6
    # two utterances, other f0 tracks (one fully unvoiced), against the oracle
    f0b = torch.tensor([[220.0, 220.0, 0.0, 5.0, 330.5, 331.0, 100.0, 9.9, 10.1, 440.0], [0.0] * 10])
    phase2, noise2 = reference_draws(7, 2, 10)
    s2 = gen.source(f0b, phase=phase2, noise=noise2).cpu()
    w64 = {k: v.double() for k, v in w.items()}
    want = HO.harmonic_source(w64, f0b, phase2, noise2, dtype=torch.float64)
    assert (s2.double() - want).abs().max().item() <= 1e-4


def test_source_of_a_long_utterance_keeps_the_references_phase(w):
    """1500 frames (30 s, 720 000 samples): the phase of the ninth harmonic runs through thousands of cycles, where the
    reference's fp32 storage of the running sum quantises it; the device reproduces that (1e-6 against the oracle in
    the reference's arithmetic; observed 1.5e-7)."""
    from chatterbox_vllm_amd.hift import HiFTGenerator
    from oracle import hift_oracle as HO
    T = 1500
    g = torch.Generator().manual_seed(1)
    f0 = (80 + 200 * torch.rand(1, T, generator=g)).float()
    f0[0, 100:130] = 0.0
    phase = (torch.rand(1, 9, 1, generator=g) * 2 - 1) * 3.14159
    noise = torch.randn(1, 9, T * 480, generator=g)
    long_gen = HiFTGenerator(HW.synthetic_state_dict(), max_batch=1, max_frames=T)
    s = long_gen.source(f0, phase=phase, noise=noise).cpu()
    long_gen.shutdown()
    want = HO.harmonic_source(w, f0, phase, noise)
    assert (s - want).abs().max().item() <= 1e-6


def test_source_drawn_on_the_device(gen, w):
    """Without draws handed in the source is a function of the seed, and its noise has the reference's statistics: in
    an unvoiced utterance the pre-tanh signal is N(0, (0.1/3)^2 * |w|^2) + bias."""
    f0 = torch.zeros(2, 40)
    f0[1] = 200.0
    a = gen.source(f0, seed=11).cpu()
    b = gen.source(f0, seed=11).cpu()
    c = gen.source(f0, seed=12).cpu()
    assert torch.equal(a, b) and not torch.equal(a, c)
    lw, lb = w["m_source.l_linear.weight"][0], w["m_source.l_linear.bias"][0]
    z = torch.atanh(a[0, 0].double().clamp(-0.999999, 0.999999)) - lb.double()
    sigma = (0.1 / 3) * lw.double().norm().item()
    assert abs(z.mean().item()) <= 4 * sigma / (z.numel() ** 0.5) + 1e-6
    assert abs(z.std().item() / sigma - 1) <= 0.03
    # voiced: the deterministic part dominates; the noise there has std 0.003 per harmonic
    v1, v2 = gen.source(f0, seed=1).cpu()[1, 0], gen.source(f0, seed=2).cpu()[1, 0]
    assert 0.0 < (v1 - v2).abs().max().item()


def test_decode_matches_the_reference(gen, fx):
    """decode() of two utterances against the reference's output: 2e-6 on a waveform of amplitude 0.12 (sums of up to
    2816 fp32 products in another order through 74 convolutions; observed 1.6e-7)."""
    T2 = int(fx["b_frames"])
    mel2 = HW.synthetic_mel(T2, seed=int(fx["b_mel_seed"]), batch=2)
    s2 = handmade_source(T2, 2, int(fx["b_source_seed"]))
    wav = gen.decode(mel2, s2).cpu()
    ref = torch.from_numpy(fx["b_wav"])
    assert wav.shape == ref.shape
    err = (wav - ref).abs().max().item()
    print("decode vs reference: max |err| %.3g on amplitude %.3g" % (err, ref.abs().max().item()))
    assert err <= 2e-6
    # one utterance alone: the same bits (a row's result does not depend on the batch)
    alone = gen.decode(mel2[1:], s2[1:]).cpu()
    assert torch.equal(alone[0], wav[1])


def test_decode_long_utterance_against_the_oracle(gen, w):
    """64 frames (30 720 samples, several tiles per launch in every stage) against the fp64 oracle."""
    from oracle import hift_oracle as HO
    T = 64
    mel = HW.synthetic_mel(T, seed=21)
    s = handmade_source(T, 1, 22)
    wav = gen.decode(mel, s).cpu()
    w64 = {k: v.double() for k, v in w.items()}
    want = HO.decode(w64, mel, s, dtype=torch.float64)
    err = (wav.double() - want).abs().max().item()
    print("decode vs fp64 oracle: max |err| %.3g on amplitude %.3g" % (err, want.abs().max().item()))
    assert err <= 2e-6
    assert wav.abs().max().item() <= 0.99


def test_inference_uses_its_own_source_and_the_cache(gen, fx):
    T = int(fx["a_frames"])
    mel = HW.synthetic_mel(T, seed=int(fx["a_mel_seed"]))
    wav, s = gen.inference(mel, seed=5)
    assert wav.shape == (1, T * 480) and s.shape == (1, 1, T * 480)
    assert torch.equal(gen.decode(mel, s), wav)
    f0 = gen.f0_predictor(mel)
    assert torch.equal(gen.source(f0, seed=5), s)
    cache = torch.from_numpy(fx["a_cache"])
    wav_c, s_c = gen.inference(mel, cache_source=cache, seed=5)
    assert torch.equal(s_c[:, :, :600].cpu(), cache) and torch.equal(s_c[:, :, 600:], s[:, :, 600:])
    assert torch.equal(gen.decode(mel, s_c), wav_c)
    # with the reference's draws handed to source(), the whole path lands on the reference's waveform
    torch.manual_seed(int(fx["a_rng_seed"]))
    phase = torch.distributions.uniform.Uniform(low=-np.pi, high=np.pi).sample(sample_shape=(1, 9, 1))
    noise = torch.randn(1, 9, T * 480)
    s_ref = gen.source(f0, phase=phase, noise=noise)
    wav_ref = gen.decode(mel, s_ref).cpu()
    assert (wav_ref - torch.from_numpy(fx["a_wav"])).abs().max().item() <= 5e-5


def test_interior_of_a_long_utterance_equals_the_interior_of_a_window(gen):
    """Size-independent property: away from the ends a sample depends only on the mel frames and source samples within
    the network's receptive field, and every output is summed in the same order wherever it falls in a tile, so the
    middle of a 300-frame utterance equals, bit for bit, the middle of a 140-frame window cut from it."""
    T, lo, hi = 300, 90, 230
    mel = HW.synthetic_mel(T, seed=31)
    s = handmade_source(T, 1, 32)
    whole = gen.decode(mel, s).cpu()
    part = gen.decode(mel[:, :, lo:hi].contiguous(), s[:, :, lo * 480: hi * 480].contiguous()).cpu()
    a, b = lo + 50, hi - 50
    assert torch.equal(whole[0, a * 480: b * 480], part[0, (a - lo) * 480: (b - lo) * 480])


def test_fused_residual_step_gives_the_same_bits(gen, monkeypatch):
    """The one-launch form of a residual-block step on the 64-channel stage (off by default: measured slower) must give
    the two-launch form's waveform bit for bit."""
    T = 40
    mel = HW.synthetic_mel(T, seed=51, batch=2)
    s = handmade_source(T, 2, 52)
    base = gen.decode(mel, s).clone()
    monkeypatch.setenv("S3HIFT_FUSED", "1")
    fused = gen.decode(mel, s).clone()
    monkeypatch.delenv("S3HIFT_FUSED")
    assert torch.equal(fused, base)


def test_workspace_limits_are_errors(gen):
    with pytest.raises(ValueError):
        gen.decode(HW.synthetic_mel(321, seed=1), torch.zeros(1, 1, 321 * 480))
    with pytest.raises(ValueError):
        gen.f0_predictor(HW.synthetic_mel(4, seed=1, batch=3))
