"""``ChatterboxTTS``: the reference's public class with the vLLM-backed T3 path replaced by
the MI355X engine.

Kept from src/chatterbox_vllm/tts.py of the reference: the constructor names
(``from_pretrained`` :160-190, ``from_pretrained_multilingual`` :192-206, ``from_local``
:91-158), ``generate`` / ``generate_with_conds`` and their defaults (:259-312),
``get_audio_conditionals`` (:215-247), ``update_exaggeration`` (:249-257),
``get_supported_languages`` (:208-213), ``sr`` (:86-89) and ``shutdown`` (:564-566); the
sampling-parameter derivation (language clamp :350-356, CHATTERBOX_DETERMINISTIC :358-362,
batch-wide max_tokens :372-393) and the id post-processing (:435-486).

Scope: ``generate`` returns one 1-D int64 tensor of speech tokens per prompt, or, when a
token -> waveform object with the reference's ``inference(speech_tokens=, ref_dict=,
n_timesteps=)`` method is attached as ``s3gen``, the waveform that returns (tts.py:489-493).
``from_local`` attaches this package's ``S3Token2Wav`` (s3gen.py) when the checkpoint
directory holds ``s3gen.safetensors``; ``from_synthetic(with_s3gen=True)`` does on seeded
weights; the waveform then gets the reference's alignment cap and RMS tail trim
(``postprocess_waveform``, tts.py:495-541).  Not part of the package: the optional VAD trim
(tts.py:543-555, torchaudio) and the audio front end behind ``audio_prompt_path``
(VoiceEncoder, S3Tokenizer, the reference-audio side of S3Gen).
"""
from __future__ import annotations

import os
import time
from functools import lru_cache
from pathlib import Path
from typing import Any, Callable, Dict, List, Optional, Sequence, Tuple, Union

import torch

from .config import (T3ModelConfig, CONDITIONING_SIZE, SPEECH_TOKEN_OFFSET, START_SPEECH_TOKEN,
                     STOP_SPEECH_TOKEN)
from .engine import T3Engine, SamplingParams, RequestOutput
from .text_utils import SUPPORTED_LANGUAGES, punc_norm, batch_max_tokens, guard_cap, _env_flag
from .weights import synthetic_state_dict, synthetic_cond_emb
from . import cond_enc as _cond_enc

REPO_ID = "ResembleAI/chatterbox"
S3GEN_SR = 24000
STOP_OFFSET_ID = STOP_SPEECH_TOKEN + SPEECH_TOKEN_OFFSET


def postprocess_speech_ids(token_ids: Sequence[int], prompt: str = "", tail_crop: Optional[int] = None) -> torch.Tensor:
    """Engine ids (+1000 space, stop id included) -> speech tokens for the vocoder.

    tts.py:448-456 cut at the first stop id and drop CHATTERBOX_TAIL_CROP_TOKENS (2) more;
    :463-465 remove the offset, keep what lies between the first SOS and the first EOS
    (models/s3tokenizer/__init__.py:37-79), keep ids < 6561; :467-486 cap at the per-prompt guard."""
    if tail_crop is None:
        tail_crop = int(os.environ.get("CHATTERBOX_TAIL_CROP_TOKENS", "2"))
    ids = list(token_ids)
    if STOP_OFFSET_ID in ids:
        ids = ids[: max(0, ids.index(STOP_OFFSET_ID) - max(0, int(tail_crop)))]
    ids = [t - SPEECH_TOKEN_OFFSET for t in ids]
    start = ids.index(START_SPEECH_TOKEN) + 1 if START_SPEECH_TOKEN in ids else 0
    end = ids.index(STOP_SPEECH_TOKEN) if STOP_SPEECH_TOKEN in ids else len(ids)
    ids = ids[start:end] if end > start else []
    ids = [t for t in ids if t < START_SPEECH_TOKEN]
    if prompt:
        ids = ids[: guard_cap(prompt)]
    return torch.tensor(ids, dtype=torch.int64)


def _s3gen_split_mode() -> int:
    """CHATTERBOX_S3GEN_SPLIT_BF16: 0 / unset = the flow-matching decoder's matrix launches on the fp32 matrix instruction (the
    reference's arithmetic); 1 = its attention on split-bf16 operands; 2 = its linear layers and convolutions too."""
    v = os.environ.get("CHATTERBOX_S3GEN_SPLIT_BF16", "0").strip().lower()
    return {"": 0, "0": 0, "false": 0, "off": 0, "1": 1, "true": 1, "on": 1, "yes": 1, "2": 2}.get(v, 0)


def _env_ms(name: str, default: str, sr: int) -> int:
    """An environment variable holding whole milliseconds, as samples at `sr`."""
    return int(sr * int(os.environ.get(name, default)) / 1000)


def _token_duration_cap(n_samples: int, n_tokens: int, sr: int) -> int:
    """Samples to keep under the alignment rule: 25 speech tokens per second, plus CHATTERBOX_ALIGN_SAFETY_MS."""
    return min(n_samples, int(round(n_tokens * (sr / 25))) + max(0, _env_ms("CHATTERBOX_ALIGN_SAFETY_MS", "0", sr)))


def _quiet_tail_cut(wav: torch.Tensor, sr: int) -> int:
    """Samples to keep under the tail rule: windowed RMS (CHATTERBOX_RMS_WINDOW_MS long, every CHATTERBOX_RMS_HOP_MS);
    the last window louder than the threshold ends the utterance, CHATTERBOX_TAIL_TRIM_SAFETY_MS later.  The threshold
    is CHATTERBOX_TAIL_TRIM_DB_REL below the loudest window, or the absolute CHATTERBOX_TAIL_TRIM_DB for pure silence."""
    n = wav.shape[1]
    window, hop = max(1, _env_ms("CHATTERBOX_RMS_WINDOW_MS", "50", sr)), max(1, _env_ms("CHATTERBOX_RMS_HOP_MS", "20", sr))
    if n < window:
        return n
    # mean square per window as a strided convolution with a box filter: the reference's own arithmetic, so a window on
    # the threshold falls on the same side
    box = torch.full((1, 1, window), 1.0 / window, device=wav.device, dtype=wav.dtype)
    level = torch.nn.functional.conv1d(wav.square().reshape(1, 1, -1), box, stride=hop).flatten().sqrt()
    loudest = float(level.max()) if level.numel() else 0.0
    db = float(os.environ.get("CHATTERBOX_TAIL_TRIM_DB_REL", "-35")) if loudest > 0 else float(os.environ.get("CHATTERBOX_TAIL_TRIM_DB", "-42"))
    floor = (loudest if loudest > 0 else 1.0) * 10.0 ** (db / 20.0)
    audible = (level > floor).nonzero()
    if audible.numel() == 0:
        return n
    return min(n, (int(audible[-1]) + 1) * hop + _env_ms("CHATTERBOX_TAIL_TRIM_SAFETY_MS", "50", sr))


def postprocess_waveform(wav: torch.Tensor, n_tokens: int, sr: int = S3GEN_SR) -> torch.Tensor:
    """What the reference does to a rendered waveform [1, N] before returning it (tts.py:495-541), same environment switches:
    cut to the duration the speech tokens stand for (CHATTERBOX_ALIGN_HARD), then cut the quiet tail
    (CHATTERBOX_TAIL_TRIM).  The optional VAD trim behind CHATTERBOX_VAD_TRIM (tts.py:543-555) needs torchaudio and is not
    reproduced."""
    if wav is None or wav.numel() == 0:
        return wav
    if _env_flag("CHATTERBOX_ALIGN_HARD", "1"):
        wav = wav[:, : _token_duration_cap(wav.shape[1], n_tokens, sr)]
    if _env_flag("CHATTERBOX_TAIL_TRIM", "1") and wav.numel() > 0:
        wav = wav[:, : _quiet_tail_cut(wav, sr)]
    return wav


class ChatterboxTTS:
    variant_config = staticmethod(lambda variant: T3ModelConfig.english() if variant == "english"
                                  else T3ModelConfig.multilingual())

    def __init__(self, target_device: str, max_model_len: int, t3: T3Engine, t3_config: T3ModelConfig,
                 cond_enc_weights: Optional[Dict[str, torch.Tensor]] = None,
                 default_conds: Optional[Dict[str, Any]] = None, s3gen: Any = None, ve: Any = None,
                 variant: str = "english", speech_emb: Optional[torch.Tensor] = None,
                 speech_pos_emb: Optional[torch.Tensor] = None):
        self.target_device = target_device
        self.max_model_len = max_model_len
        self.t3 = t3
        self.t3_config = t3_config
        self.cond_enc_weights = cond_enc_weights
        self.default_conds = default_conds
        self.s3gen = s3gen
        self.ve = ve
        self.variant = variant
        self._speech_emb = speech_emb
        self._speech_pos_emb = speech_pos_emb
        self._synthetic_voices = default_conds is None
        self._cond_encoder: Optional[_cond_enc.CondEncoder] = None
        self.last_t3_seconds = 0.0
        self.last_request_outputs: List[RequestOutput] = []

    @property
    def sr(self) -> int:
        return S3GEN_SR

    # ---- construction ------------------------------------------------------------------
    @classmethod
    def _build(cls, state_dict, cfg: T3ModelConfig, tokenizer, *, target_device="cuda", max_model_len=1000,
               compile=False, max_batch_size=10, variant="english", default_conds=None, s3gen=None, ve=None,
               **kwargs) -> "ChatterboxTTS":
        device = 0
        if isinstance(target_device, str) and ":" in target_device:
            device = int(target_device.split(":")[1])
        # `compile` selected CUDA graphs in the reference (enforce_eager = not compile, tts.py:136); graph
        # replay costs nothing here, so it stays on unless enforce_eager is passed explicitly.
        kwargs.setdefault("enforce_eager", False)
        kwargs.pop("gpu_memory_utilization", None)            # the KV pool is sized from max_batch_size directly
        max_num_seqs = int(kwargs.pop("max_num_seqs", max_batch_size))
        engine = T3Engine(cfg, state_dict, tokenizer, max_model_len=max_model_len, max_num_seqs=max_num_seqs,
                          device=device, **kwargs)
        ce = {k[len("cond_enc."):]: v.float() for k, v in state_dict.items() if k.startswith("cond_enc.")}
        return cls(target_device, max_model_len, engine, cfg, cond_enc_weights=ce or None,
                   default_conds=default_conds, s3gen=s3gen, ve=ve, variant=variant,
                   speech_emb=state_dict["speech_emb.weight"].float(),
                   speech_pos_emb=state_dict["speech_pos_emb.emb.weight"].float())

    @classmethod
    def from_local(cls, ckpt_dir: Union[str, Path], target_device: str = "cuda", max_model_len: int = 1000,
                   compile: bool = False, max_batch_size: int = 10, variant: str = "english",
                   s3gen_use_fp16: bool = False, **kwargs) -> "ChatterboxTTS":
        from safetensors.torch import load_file
        from .tokenizer import EnTokenizer, MTLTokenizer
        ckpt_dir = Path(ckpt_dir)
        names = ["t3_cfg.safetensors"] if variant == "english" else ["t3_mtl23ls_v2.safetensors", "t3_23lang.safetensors"]
        path = next((ckpt_dir / n for n in names if (ckpt_dir / n).is_file()), None)
        if path is None:
            raise FileNotFoundError(f"no T3 checkpoint ({' / '.join(names)}) in {ckpt_dir}")
        sd = load_file(str(path))
        tok = (EnTokenizer if variant == "english" else MTLTokenizer).from_pretrained(str(ckpt_dir))
        conds = None
        if (ckpt_dir / "conds.pt").is_file():
            conds = torch.load(ckpt_dir / "conds.pt", weights_only=True, map_location="cpu")
        if kwargs.get("s3gen") is None and (ckpt_dir / "s3gen.safetensors").is_file():
            # the reference's S3Gen checkpoint (tts.py:146-148): its flow.* and mel2wav.* entries feed the token -> waveform
            # chain of this package; the reference-audio side of S3Gen (tokenizer, speaker encoder) is not built
            from .s3gen import S3Token2Wav
            s3_sd = load_file(str(ckpt_dir / "s3gen.safetensors"))
            kwargs["s3gen"] = S3Token2Wav({k[len("flow."):]: v for k, v in s3_sd.items() if k.startswith("flow.")},
                                          {k[len("mel2wav."):]: v for k, v in s3_sd.items() if k.startswith("mel2wav.")},
                                          max_batch=min(4, max(1, max_batch_size)), chains=2,
                                          split_bf16=_s3gen_split_mode())
        cfg = cls.variant_config(variant)
        # the backbone's depth is the checkpoint's (30 for the published ones): a cut-down checkpoint loads as what it is
        depth = 1 + max((int(k.split(".")[2]) for k in sd if k.startswith("tfmr.layers.")), default=-1)
        if depth != cfg.n_layers:
            cfg = T3ModelConfig(n_layers=depth, text_vocab=cfg.text_vocab)
        return cls._build(sd, cfg, tok, target_device=target_device,
                          max_model_len=max_model_len, compile=compile, max_batch_size=max_batch_size,
                          variant=variant, default_conds=conds, **kwargs)

    @classmethod
    def from_pretrained(cls, repo_id: str = REPO_ID, revision: str = "1b475dffa71fb191cb6d5901215eb6f55635a9b6",
                        *args, **kwargs) -> "ChatterboxTTS":
        from huggingface_hub import hf_hub_download
        local = None
        for f in ["t3_cfg.safetensors", "tokenizer.json", "conds.pt"]:
            local = hf_hub_download(repo_id=repo_id, filename=f, revision=revision)
        return cls.from_local(Path(local).parent, *args, variant="english", **kwargs)

    @classmethod
    def from_pretrained_multilingual(cls, repo_id: str = REPO_ID, revision: str = "main", *args, **kwargs):
        from huggingface_hub import hf_hub_download
        local = None
        for f in ["t3_mtl23ls_v2.safetensors", "grapheme_mtl_merged_expanded_v1.json", "conds.pt", "Cangjie5_TC.json"]:
            local = hf_hub_download(repo_id=repo_id, filename=f, revision=revision)
        return cls.from_local(Path(local).parent, *args, variant="multilingual", **kwargs)

    @classmethod
    def from_synthetic(cls, variant: str = "english", tokenizer: Optional[Callable] = None, seed: int = 1234,
                       n_layers: Optional[int] = None, with_s3gen: bool = False, **kwargs) -> "ChatterboxTTS":
        """Seeded random checkpoint of the exact architecture (no network needed): what the
        benchmark and the tests run on.  Voices are seeded conditioning prefixes.  ``with_s3gen`` attaches the token ->
        waveform chain on seeded weights too, so that ``generate`` returns waveforms as the reference's does."""
        if with_s3gen and kwargs.get("s3gen") is None:
            from . import cfm_weights, flow_weights, hift_weights
            from .s3gen import S3Token2Wav
            flow_sd = dict(flow_weights.synthetic_state_dict())
            flow_sd.update(cfm_weights.synthetic_state_dict())
            kwargs["s3gen"] = S3Token2Wav(flow_sd, hift_weights.synthetic_state_dict(), max_tokens=kwargs.get("max_model_len", 1000) + 256,
                                          max_batch=min(4, max(1, kwargs.get("max_batch_size", 10))), chains=2,
                                          split_bf16=_s3gen_split_mode())
        cfg = cls.variant_config(variant)
        if n_layers is not None:
            cfg = T3ModelConfig(n_layers=n_layers, text_vocab=cfg.text_vocab)
        sd = synthetic_state_dict(cfg, seed)
        sd.update(_cond_enc.synthetic_weights(seed))
        return cls._build(sd, cfg, tokenizer, variant=variant, **kwargs)

    # ---- conditioning ---------------------------------------------------------------------
    def _encoder(self) -> "_cond_enc.CondEncoder":
        if self.cond_enc_weights is None:
            raise RuntimeError("checkpoint has no cond_enc.* weights")
        if self._cond_encoder is None:
            self._cond_encoder = _cond_enc.CondEncoder(self.cond_enc_weights, device=self.t3.device)
        return self._cond_encoder

    def get_supported_languages(self) -> Dict[str, str]:
        return SUPPORTED_LANGUAGES.copy() if self.variant == "multilingual" else {"en": "English"}

    @lru_cache(maxsize=10)
    def get_audio_conditionals(self, wav_fpath: Optional[str] = None) -> Tuple[Dict[str, Any], torch.Tensor]:
        """(vocoder reference dict, cond_emb fp32 [34, 1024] on the CPU)."""
        if wav_fpath is not None and self.ve is None:
            if self._synthetic_voices:
                voice = sum(wav_fpath.encode()) % 997 + 1          # a stable seeded voice per path
                return self._synthetic_ref(voice), synthetic_cond_emb(voice, self.t3_config.d_model)
            raise NotImplementedError("audio_prompt_path needs the audio front end (voice encoder + S3 tokenizer), "
                                      "which is outside this package; pass cond_emb to generate_with_conds")
        if self.default_conds is None:
            return self._synthetic_ref(0), synthetic_cond_emb(0, self.t3_config.d_model)
        t3c = self.default_conds["t3"]
        tokens = torch.atleast_2d(t3c["cond_prompt_speech_tokens"]).long()
        emb = self._speech_emb[tokens[0]] + self._speech_pos_emb[: tokens.shape[1]]       # tts.py:238
        cond = self._encoder().encode(t3c["speaker_emb"].float(), emb, 0.5)
        return self.default_conds.get("gen", {}), cond.to("cpu")

    def _synthetic_ref(self, voice: int) -> Dict[str, Any]:
        """A seeded vocoder reference (the `gen` half of the reference's Conditionals, s3gen.py:120-170: prompt tokens, their
        mel frames, x-vector) for a seeded voice; empty when no token -> waveform chain is attached."""
        if self.s3gen is None:
            return {}
        g = torch.Generator().manual_seed(7000 + voice)
        n = 50                                                     # two seconds of prompt
        return {"prompt_token": torch.randint(0, 6561, (1, n), generator=g), "prompt_token_len": torch.tensor([n]),
                "prompt_feat": 2.0 * torch.randn(1, 2 * n, 80, generator=g) - 4.0, "prompt_feat_len": None,
                "embedding": torch.randn(1, 192, generator=g)}

    def update_exaggeration(self, cond_emb: torch.Tensor, exaggeration: float) -> torch.Tensor:
        if exaggeration == 0.5:
            return cond_emb
        out = cond_emb.clone()
        out[-1] = self._encoder().emotion_row(exaggeration)
        return out

    # ---- generation -----------------------------------------------------------------------
    def generate(self, prompts: Union[str, List[str]], audio_prompt_path: Optional[str] = None,
                 language_id: Optional[str] = "en", exaggeration: float = 0.5, temperature: float = 0.8,
                 max_tokens: int = 1000, top_p: float = 0.8, repetition_penalty: float = 2.0, *args, **kwargs):
        s3gen_ref, cond_emb = self.get_audio_conditionals(audio_prompt_path)
        return self.generate_with_conds(prompts=prompts, s3gen_ref=s3gen_ref, cond_emb=cond_emb,
                                        temperature=temperature, language_id=language_id, exaggeration=exaggeration,
                                        max_tokens=max_tokens, top_p=top_p, repetition_penalty=repetition_penalty,
                                        *args, **kwargs)

    def _request_items(self, prompts: List[str], language_id, cond_emb) -> List[dict]:
        texts = ["[START]" + punc_norm(p) + "[STOP]" for p in prompts]                    # tts.py:327
        if self.variant == "multilingual":
            # the multilingual route of the reference is mtl_tts.py:419-435: "[xx]" tag + tokenization hint
            items = []
            for t in texts:
                it = {"prompt": (f"[{language_id.lower()}]" + t) if language_id else t,
                      "multi_modal_data": {"conditionals": [cond_emb]}}
                if language_id:
                    it["tokenization_kwargs"] = {"language_id": language_id.lower()}
                items.append(it)
            return items
        return [{"prompt": t, "multi_modal_data": {"conditionals": [cond_emb]}} for t in texts]

    def generate_with_conds(self, prompts: Union[str, List[str]], s3gen_ref: Dict[str, Any], cond_emb: torch.Tensor,
                            language_id: Optional[str] = "en", temperature: float = 0.8, exaggeration: float = 0.5,
                            max_tokens: int = 1000, diffusion_steps: int = 10, top_p: float = 1.0,
                            min_p: float = 0.05, repetition_penalty: float = 2.0, *args, **kwargs):
        if isinstance(prompts, str):
            prompts = [prompts]
        if language_id and language_id.lower() not in self.get_supported_languages():
            raise ValueError(f"Unsupported language_id '{language_id}'. Supported languages: "
                             + ", ".join(self.get_supported_languages().keys()))
        cond_emb = self.update_exaggeration(cond_emb, exaggeration)
        items = self._request_items(prompts, language_id, cond_emb)
        tokens_only = kwargs.pop("tokens_only", False)             # package extension: stop after the T3 half (see _speech_tokens)
        toks = self._speech_tokens(items, bool(language_id), temperature, top_p, max_tokens, repetition_penalty,
                                   *args, **kwargs)
        return toks if self.s3gen is None or tokens_only else self.render_waveforms(toks, s3gen_ref, diffusion_steps)

    def _speech_tokens(self, items: List[dict], clamp: bool, temperature: float, top_p: float, max_tokens: int,
                       repetition_penalty: float, *args, per_prompt_max_tokens: bool = False, **kwargs) -> List[torch.Tensor]:
        """Request items -> one tensor of speech tokens each (the T3 half of the reference's ``generate_with_conds``).

        ``per_prompt_max_tokens`` has no counterpart in the reference, whose calls share one ``max_tokens`` (the smallest
        prompt's, tts.py:378-393): with it every prompt gets the cap it would get in a call of its own, which is what lets
        the streaming server take the chunks of a request through one batched call and still return what chunk-by-chunk
        calls return.  ``clamp`` (the reference's temperature / top_p clamp when a language id is given) may be a list
        with one entry per prompt, for the same reason: chunks of one request may resolve to different languages."""
        prompts = [it["prompt"] for it in items]
        clamps = [bool(c) for c in clamp] if isinstance(clamp, (list, tuple)) else [bool(clamp)] * len(prompts)

        # NOTE the reference forwards min_p only when the caller passes it through **kwargs (tts.py:410-417):
        # the named `min_p` parameter is accepted and not used.
        def params(group, clamped):
            temp_use, top_p_use = temperature, top_p
            if clamped:                                                # tts.py:350-356
                temp_use, top_p_use = min(temperature, 0.5), min(top_p, 0.5)
            if _env_flag("CHATTERBOX_DETERMINISTIC"):                  # tts.py:358-362
                temp_use, top_p_use = 0.0, 1.0
            return SamplingParams(temperature=temp_use, stop_token_ids=[STOP_OFFSET_ID],
                                  max_tokens=batch_max_tokens(group, max_tokens, self.max_model_len), top_p=top_p_use,
                                  repetition_penalty=repetition_penalty, *args, **kwargs)
        if per_prompt_max_tokens or len(set(clamps)) > 1:
            sps = [params([p] if per_prompt_max_tokens else prompts, c) for p, c in zip(prompts, clamps)]
        else:
            sps = params(prompts, bool(clamps) and clamps[0])
        t0 = time.time()
        outs = self.t3.generate(items, sampling_params=sps)
        self.last_t3_seconds = time.time() - t0
        self.last_request_outputs = outs
        print(f"[T3] Speech Token Generation time: {self.last_t3_seconds:.2f}s")
        return [postprocess_speech_ids(out.token_ids, it["prompt"]) for it, ro in zip(items, outs) for out in ro.outputs]

    def render_waveforms(self, all_toks: List[torch.Tensor], s3gen_ref: Dict[str, Any], diffusion_steps: int = 10) -> List[torch.Tensor]:
        """Speech tokens -> waveforms [1, N] on the CPU through the attached token -> waveform object, each with the
        reference's alignment cap and RMS tail trim (tts.py:489-541)."""
        results: List[Any] = [None] * len(all_toks)
        if hasattr(self.s3gen, "render_many"):
            # the reference takes the utterances through S3Gen one after the other (tts.py:489-493); this package's chain
            # takes those of similar length through the flow-matching stage together and keeps two copies of itself busy
            wavs = self.s3gen.render_many([t.to(self.target_device) for t in all_toks], s3gen_ref, n_timesteps=diffusion_steps)
            for i, wav in enumerate(wavs):
                if wav is not None:
                    results[i] = postprocess_waveform(wav, int(all_toks[i].numel()), self.sr).cpu()
        for i, toks in enumerate(all_toks):
            if results[i] is None:
                wav, _ = self.s3gen.inference(speech_tokens=toks.to(self.target_device), ref_dict=s3gen_ref,
                                              n_timesteps=diffusion_steps)
                results[i] = postprocess_waveform(wav, int(toks.numel()), self.sr).cpu()
        return results

    def shutdown(self):
        if self._cond_encoder is not None:
            self._cond_encoder.close()
            self._cond_encoder = None
        if self.t3 is not None:
            self.t3.shutdown()
        self.t3 = None
        if self.s3gen is not None and hasattr(self.s3gen, "shutdown"):
            self.s3gen.shutdown()
            self.s3gen = None
