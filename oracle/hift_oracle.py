"""CPU restatement of the reference's HiFT vocoder (mel -> waveform).  TEST INFRASTRUCTURE ONLY: imported by tests/,
by the golden-vector scripts and by bench.py's cpu_baseline leg, never by the package.

Follows /root/reference/src/chatterbox_vllm/models/s3gen/hifigan.py (HiFTGenerator with the arguments of
s3gen.py:237-245: 24 kHz, upsampling 8 x 5 x 3, ISTFT n_fft 16 / hop 4) and f0_predictor.py, written as plain functions
over a dict of tensors instead of nn.Modules, with the STFT and the inverse STFT spelled out as sums (torch.stft /
torch.istft are not called), so that every index rule the HIP kernels must reproduce is visible here.

Pinned by tests/golden/hift_fixture.npz: outputs of the reference's own classes (its two source files loaded by path,
the package's seeded synthetic weights loaded into them) for f0, the harmonic source with the reference's random draws,
and decode(); tests/test_hift_oracle.py holds this file to them.

All functions take `dtype` (torch.float32 as the reference computes, or float64 to measure rounding)."""
import math

import torch
import torch.nn.functional as F

SR = 24000
UP_RATES = (8, 5, 3)
UP_KERNELS = (16, 11, 7)
RES_KERNELS = (3, 7, 11)
RES_DILATIONS = (1, 3, 5)
SRC_RES_KERNELS = (7, 7, 11)
N_FFT, HOP = 16, 4
NB_HARMONICS = 8
SINE_AMP, NOISE_STD, VOICED_THRESHOLD = 0.1, 0.003, 10.0
LRELU_SLOPE = 0.1
AUDIO_LIMIT = 0.99
SAMPLES_PER_FRAME = 8 * 5 * 3 * HOP            # 480 waveform samples per mel frame


def fold_weight_norm(sd, dtype=torch.float32):
    """weight_norm parametrisation -> plain weights: w = g * v / ||v||, the norm taken over all dims but 0
    (torch.nn.utils.parametrizations.weight_norm with its default dim=0, as hifigan.py:120, 331, 339 and
    f0_predictor.py:29 apply it; for ConvTranspose1d dim 0 is the INPUT channel)."""
    out = {}
    for k, v in sd.items():
        if k.endswith(".parametrizations.weight.original1"):
            base = k[: -len(".parametrizations.weight.original1")]
            g = sd[base + ".parametrizations.weight.original0"].to(dtype)
            vv = v.to(dtype)
            norm = vv.flatten(1).norm(dim=1).view(-1, *([1] * (vv.dim() - 1)))
            out[base + ".weight"] = g * vv / norm
        elif k.endswith(".parametrizations.weight.original0"):
            continue
        else:
            out[k] = v.to(dtype)
    return out


def snake(x, alpha):
    """hifigan.py:73-84: x + 1/(alpha + 1e-9) * sin(alpha x)^2, alpha per channel."""
    a = alpha.view(1, -1, 1)
    return x + (1.0 / (a + 1e-9)) * torch.sin(x * a) ** 2


def resblock(w, prefix, x, kernel):
    """hifigan.py:154-161."""
    for i, d in enumerate(RES_DILATIONS):
        xt = snake(x, w[f"{prefix}.activations1.{i}.alpha"])
        xt = F.conv1d(xt, w[f"{prefix}.convs1.{i}.weight"], w[f"{prefix}.convs1.{i}.bias"], dilation=d,
                      padding=(kernel * d - d) // 2)
        xt = snake(xt, w[f"{prefix}.activations2.{i}.alpha"])
        xt = F.conv1d(xt, w[f"{prefix}.convs2.{i}.weight"], w[f"{prefix}.convs2.{i}.bias"], padding=(kernel - 1) // 2)
        x = xt + x
    return x


def f0_predict(w, mel):
    """f0_predictor.py:52-55: five (conv k3 + ELU), a linear layer to one value per frame, abs.  mel [B, 80, T]."""
    x = mel
    for i in (0, 2, 4, 6, 8):
        x = F.elu(F.conv1d(x, w[f"f0_predictor.condnet.{i}.weight"], w[f"f0_predictor.condnet.{i}.bias"], padding=1))
    y = x.transpose(1, 2) @ w["f0_predictor.classifier.weight"].t() + w["f0_predictor.classifier.bias"]
    return y.squeeze(-1).abs()


def hann_window(dtype):
    """scipy.signal.get_window("hann", 16, fftbins=True) (hifigan.py:379): the periodic Hann window."""
    n = torch.arange(N_FFT, dtype=torch.float64)
    return (0.5 - 0.5 * torch.cos(2 * math.pi * n / N_FFT)).to(torch.float32).to(dtype)


def harmonic_source(w, f0, phase_vec, noise, dtype=torch.float32):
    """hifigan.py:455-457 + SourceModuleHnNSF.forward :267-283 + SineGen.forward :201-231.
    f0 [B, T] (Hz per frame); phase_vec [B, 9, 1] (harmonic 0's entry is ignored: the reference zeroes it);
    noise [B, 9, 480 T] standard normal.  Returns s [B, 1, 480 T]."""
    B, T = f0.shape
    f0u = f0.to(dtype).repeat_interleave(SAMPLES_PER_FRAME, dim=1).unsqueeze(1)          # nearest upsample, [B, 1, L]
    mult = torch.arange(1, NB_HARMONICS + 2, dtype=dtype).view(1, -1, 1)
    f_mat = f0u * mult / SR                                                               # [B, 9, L]
    theta = 2 * math.pi * (torch.cumsum(f_mat, dim=-1) % 1)
    ph = phase_vec.to(dtype).clone()
    ph[:, 0, :] = 0
    sine = SINE_AMP * torch.sin(theta + ph)
    uv = (f0u > VOICED_THRESHOLD).to(dtype)
    noise_amp = uv * NOISE_STD + (1 - uv) * SINE_AMP / 3
    sine = sine * uv + noise_amp * noise.to(dtype)
    merged = torch.tanh(sine.transpose(1, 2) @ w["m_source.l_linear.weight"].t() + w["m_source.l_linear.bias"])
    return merged.transpose(1, 2)


def stft(s, dtype):
    """torch.stft(x, 16, 4, 16, window, center=True, pad_mode="reflect", return_complex) as hifigan.py:396-402 calls it:
    frame m covers padded samples 4m .. 4m+15, the padding is 8 reflected samples on both sides, and
    X[f, m] = sum_n w[n] x[4m + n] exp(-2 pi i f n / 16), f = 0..8.  s [B, L] -> real, imag [B, 9, L/4 + 1]."""
    B, L = s.shape
    xp = F.pad(s.unsqueeze(1), (N_FFT // 2, N_FFT // 2), mode="reflect").squeeze(1)
    frames = xp.unfold(1, N_FFT, HOP) * hann_window(dtype)                                # [B, M, 16]
    n = torch.arange(N_FFT, dtype=torch.float64)
    f = torch.arange(N_FFT // 2 + 1, dtype=torch.float64)
    ang = 2 * math.pi * f[:, None] * n[None, :] / N_FFT
    re = frames @ torch.cos(ang).to(dtype).t()
    im = -(frames @ torch.sin(ang).to(dtype).t())
    return re.transpose(1, 2), im.transpose(1, 2)


def istft(magnitude, phase, dtype):
    """hifigan.py:404-410: clip the magnitude at 100, X = mag (cos(phase) + i sin(phase)), then
    torch.istft(X, 16, 4, 16, window, center=True): per frame the inverse real DFT (the imaginary parts of bins 0 and 8
    do not enter), times the window, overlap-added; divided by the overlap-added squared window; 8 samples cut from
    both ends.  [B, 9, M] -> [B, 4 (M - 1)]."""
    magnitude = torch.clip(magnitude, max=1e2)
    re, im = magnitude * torch.cos(phase), magnitude * torch.sin(phase)
    B, _, M = re.shape
    n = torch.arange(N_FFT, dtype=torch.float64)
    f = torch.arange(N_FFT // 2 + 1, dtype=torch.float64)
    ang = 2 * math.pi * f[:, None] * n[None, :] / N_FFT                                   # [9, 16]
    wgt = torch.full((N_FFT // 2 + 1, 1), 2.0, dtype=torch.float64)
    wgt[0] = 1.0
    wgt[-1] = 1.0
    c = (wgt * torch.cos(ang) / N_FFT).to(dtype)
    sn = (wgt * torch.sin(ang) / N_FFT).to(dtype)
    sn[0] = 0
    sn[-1] = 0
    frames = re.transpose(1, 2) @ c - im.transpose(1, 2) @ sn                             # [B, M, 16]
    win = hann_window(dtype)
    frames = frames * win
    L = N_FFT + HOP * (M - 1)
    y = torch.zeros(B, L, dtype=dtype)
    env = torch.zeros(L, dtype=dtype)
    for m in range(M):
        y[:, HOP * m: HOP * m + N_FFT] += frames[:, m]
        env[HOP * m: HOP * m + N_FFT] += win * win
    y = y[:, N_FFT // 2: L - N_FFT // 2]
    env = env[N_FFT // 2: L - N_FFT // 2]
    return y / env


def decode(w, mel, s, dtype=torch.float32, taps=None):
    """hifigan.py:412-444.  mel [B, 80, T], s [B, 1, 480 T] -> waveform [B, 480 T].  `taps`, when a dict, receives the
    intermediate tensors by name (for per-stage comparisons)."""
    mel, s = mel.to(dtype), s.to(dtype)
    s_re, s_im = stft(s.squeeze(1), dtype)
    s_stft = torch.cat([s_re, s_im], dim=1)
    x = F.conv1d(mel, w["conv_pre.weight"], w["conv_pre.bias"], padding=3)
    if taps is not None:
        taps["s_stft"], taps["conv_pre"] = s_stft, x
    down_rates = (15, 3, 1)
    for i, (u, k) in enumerate(zip(UP_RATES, UP_KERNELS)):
        x = F.leaky_relu(x, LRELU_SLOPE)
        x = F.conv_transpose1d(x, w[f"ups.{i}.weight"], w[f"ups.{i}.bias"], stride=u, padding=(k - u) // 2)
        if i == len(UP_RATES) - 1:
            x = F.pad(x, (1, 0), mode="reflect")
        d = down_rates[i]
        if d == 1:
            si = F.conv1d(s_stft, w[f"source_downs.{i}.weight"], w[f"source_downs.{i}.bias"])
        else:
            si = F.conv1d(s_stft, w[f"source_downs.{i}.weight"], w[f"source_downs.{i}.bias"], stride=d, padding=d // 2)
        si = resblock(w, f"source_resblocks.{i}", si, SRC_RES_KERNELS[i])
        x = x + si
        if taps is not None:
            taps[f"fused.{i}"] = x
        xs = None
        for j, rk in enumerate(RES_KERNELS):
            r = resblock(w, f"resblocks.{i * 3 + j}", x, rk)
            xs = r if xs is None else xs + r
        x = xs / 3
        if taps is not None:
            taps[f"stage.{i}"] = x
    x = F.leaky_relu(x)                                    # default slope 0.01 (hifigan.py:437)
    x = F.conv1d(x, w["conv_post.weight"], w["conv_post.bias"], padding=3)
    if taps is not None:
        taps["conv_post"] = x
    magnitude = torch.exp(x[:, : N_FFT // 2 + 1])
    phase = torch.sin(x[:, N_FFT // 2 + 1:])
    y = istft(magnitude, phase, dtype)
    return torch.clamp(y, -AUDIO_LIMIT, AUDIO_LIMIT)


def inference(w, mel, phase_vec, noise, cache_source=None, dtype=torch.float32):
    """HiFTGenerator.inference (hifigan.py:462-474) with the random draws passed in."""
    f0 = f0_predict(w, mel.to(dtype))
    s = harmonic_source(w, f0, phase_vec, noise, dtype)
    if cache_source is not None and cache_source.shape[2] != 0:
        s[:, :, : cache_source.shape[2]] = cache_source.to(dtype)
    return decode(w, mel, s, dtype), s
