#!/usr/bin/env python3
"""Writes tests/golden/hift_fixture.npz: outputs of the REFERENCE's HiFT vocoder classes on seeded inputs.

Run in the build container only (it reads /root/reference):   python3 tests/golden/make_hift_fixtures.py

The reference's package cannot be imported as a package here (its s3gen/__init__.py pulls torchaudio, omegaconf,
diffusers and conformer, none installed), but the two files that make up the vocoder, hifigan.py and f0_predictor.py,
need only torch, numpy and scipy.  They are loaded by path, unmodified, the HiFTGenerator is built with the arguments of
s3gen.py:237-245, and the package's seeded synthetic weights (hift_weights.synthetic_state_dict) are loaded into it with
load_state_dict(strict=True).  Recorded: f0 per frame, the harmonic source under torch.manual_seed (the reference draws
its phases and noise from torch's global generator), decode() on that source and on a hand-made one, and inference()."""
import importlib.util
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from chatterbox_vllm_amd import hift_weights as HW     # noqa: E402

REF = "/root/reference/src/chatterbox_vllm/models/s3gen/"


def load_by_path(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def handmade_source(n_frames, batch, seed):
    """A source signal that is not the model's own: chirp + noise, amplitude of the order tanh() gives."""
    g = torch.Generator().manual_seed(seed)
    L = n_frames * HW.SAMPLES_PER_FRAME
    t = torch.arange(L, dtype=torch.float32) / HW.SR
    s = 0.6 * torch.sin(2 * torch.pi * (110.0 + 400.0 * t) * t) + 0.05 * torch.randn(batch, L, generator=g)
    return s.view(batch, 1, L).contiguous()


def main():
    torch.set_num_threads(8)
    hifigan = load_by_path("ref_hifigan", REF + "hifigan.py")
    f0mod = load_by_path("ref_f0_predictor", REF + "f0_predictor.py")
    gen = hifigan.HiFTGenerator(sampling_rate=HW.SR, upsample_rates=[8, 5, 3], upsample_kernel_sizes=[16, 11, 7],
                                source_resblock_kernel_sizes=[7, 7, 11],
                                source_resblock_dilation_sizes=[[1, 3, 5], [1, 3, 5], [1, 3, 5]],
                                f0_predictor=f0mod.ConvRNNF0Predictor())
    sd = HW.synthetic_state_dict()
    assert list(sd) == list(gen.state_dict()), "inventory differs from the reference's state dict"
    gen.load_state_dict(sd, strict=True)
    gen.eval()
    out = {}
    with torch.inference_mode():
        # case A: one utterance of 10 frames, the whole inference path
        T = 10
        mel = HW.synthetic_mel(T, seed=5)
        f0 = gen.f0_predictor(mel)
        torch.manual_seed(2024)
        wav, s = gen.inference(speech_feat=mel, cache_source=torch.zeros(1, 1, 0))
        out["a_frames"], out["a_mel_seed"], out["a_rng_seed"] = np.int64(T), np.int64(5), np.int64(2024)
        out["a_f0"], out["a_source"], out["a_wav"] = f0.numpy(), s.numpy(), wav.numpy()
        # the same with the first 600 samples of the source replaced (streaming hand-over, hifigan.py:470-472)
        cache = handmade_source(2, 1, 9)[:, :, :600]
        torch.manual_seed(2024)
        wav_c, s_c = gen.inference(speech_feat=mel, cache_source=cache)
        out["a_cache"], out["a_wav_cached"] = cache.numpy(), wav_c.numpy()
        # case B: decode() of two utterances of 7 frames on a hand-made source
        T2 = 7
        mel2 = HW.synthetic_mel(T2, seed=6, batch=2)
        s2 = handmade_source(T2, 2, 8)
        out["b_frames"], out["b_mel_seed"], out["b_source_seed"] = np.int64(T2), np.int64(6), np.int64(8)
        out["b_wav"] = gen.decode(x=mel2, s=s2).numpy()
        # intermediate tensors of case B through forward hooks (conv_pre output, conv_post output)
        taps = {}
        h1 = gen.conv_pre.register_forward_hook(lambda m, i, o: taps.__setitem__("conv_pre", o.detach().clone()))
        h2 = gen.conv_post.register_forward_hook(lambda m, i, o: taps.__setitem__("conv_post", o.detach().clone()))
        gen.decode(x=mel2, s=s2)
        h1.remove()
        h2.remove()
        out["b_conv_pre"], out["b_conv_post"] = taps["conv_pre"].numpy(), taps["conv_post"].numpy()
        re, im = gen._stft(s2.squeeze(1))
        out["b_stft_re"], out["b_stft_im"] = re.numpy(), im.numpy()
    for k, v in out.items():
        if isinstance(v, np.ndarray) and v.dtype == np.float32:
            print(k, v.shape, "absmax %.4f" % np.abs(v).max(), "mean|.| %.4f" % np.abs(v).mean())
    path = os.path.join(ROOT, "tests", "golden", "hift_fixture.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
