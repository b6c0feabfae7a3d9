"""``ChatterboxMultilingualTTS``: the 23-language entry point of the reference
(src/chatterbox_vllm/mtl_tts.py:116-667) on the MI355X engine.

Differences from ``ChatterboxTTS`` that the reference encodes and this class keeps:
per-prompt ``language_id`` lists with a count check (:405-413), the "[xx][START]...[STOP]"
prompt plus ``tokenization_kwargs`` hint (:419-435), the sampling clamp applied when any
language id is given (:453-459), defaults temperature 0.4 / top_p 0.4 (:323-328), no
language restriction to the checkpoint variant (:117-119, :351-363).
"""
from __future__ import annotations

from pathlib import Path
from typing import Any, Dict, List, Optional, Union

import torch

from .text_utils import SUPPORTED_LANGUAGES, punc_norm
from .tts import ChatterboxTTS, REPO_ID


class ChatterboxMultilingualTTS(ChatterboxTTS):
    @classmethod
    def get_supported_languages(cls) -> Dict[str, str]:
        return SUPPORTED_LANGUAGES.copy()

    @classmethod
    def from_local(cls, ckpt_dir: Union[str, Path], target_device: str = "cuda", max_model_len: int = 1000,
                   compile: bool = False, max_batch_size: int = 10, s3gen_use_fp16: bool = False, **kwargs):
        kwargs.pop("variant", None)
        return super().from_local(ckpt_dir, target_device=target_device, max_model_len=max_model_len, compile=compile,
                                  max_batch_size=max_batch_size, variant="multilingual", **kwargs)

    @classmethod
    def from_pretrained(cls, repo_id: str = REPO_ID, revision: str = "main", *args, **kwargs):
        from huggingface_hub import snapshot_download
        ckpt_dir = snapshot_download(repo_id=repo_id, repo_type="model", revision=revision, allow_patterns=[
            "t3_mtl23ls_v2.safetensors", "grapheme_mtl_merged_expanded_v1.json", "conds.pt", "Cangjie5_TC.json"])
        return cls.from_local(ckpt_dir, *args, **kwargs)

    @classmethod
    def from_synthetic(cls, tokenizer=None, seed: int = 1234, n_layers: Optional[int] = None, **kwargs):
        kwargs.pop("variant", None)
        return super().from_synthetic(variant="multilingual", tokenizer=tokenizer, seed=seed, n_layers=n_layers, **kwargs)

    def generate(self, prompts: Union[str, List[str]], language_id: Optional[Union[str, List[str]]] = None,
                 audio_prompt_path: Optional[str] = None, exaggeration: float = 0.5, temperature: float = 0.4,
                 max_tokens: int = 1000, top_p: float = 0.4, repetition_penalty: float = 2.0, *args, **kwargs):
        if language_id is not None:
            for lang in ([language_id] if isinstance(language_id, str) else language_id):
                if lang and lang.lower() not in SUPPORTED_LANGUAGES:
                    raise ValueError(f"Unsupported language_id '{lang}'. Supported languages: "
                                     + ", ".join(SUPPORTED_LANGUAGES.keys()))
        s3gen_ref, cond_emb = self.get_audio_conditionals(audio_prompt_path)
        return self.generate_with_conds(prompts=prompts, language_id=language_id, s3gen_ref=s3gen_ref,
                                        cond_emb=cond_emb, temperature=temperature, exaggeration=exaggeration,
                                        max_tokens=max_tokens, top_p=top_p, repetition_penalty=repetition_penalty,
                                        *args, **kwargs)

    def generate_with_conds(self, prompts: Union[str, List[str]], s3gen_ref: Dict[str, Any], cond_emb: torch.Tensor,
                            language_id: Optional[Union[str, List[str]]] = None, temperature: float = 0.4,
                            exaggeration: float = 0.5, max_tokens: int = 1000, diffusion_steps: int = 10,
                            top_p: float = 0.4, repetition_penalty: float = 2.0, *args, **kwargs):
        if isinstance(prompts, str):
            prompts = [prompts]
        if language_id is None:
            langs: List[Optional[str]] = [None] * len(prompts)
        elif isinstance(language_id, str):
            langs = [language_id] * len(prompts)
        else:
            langs = list(language_id)
            if len(langs) != len(prompts):
                raise ValueError(f"Number of language_ids ({len(langs)}) must match number of prompts ({len(prompts)})")
        cond_emb = self.update_exaggeration(cond_emb, exaggeration)
        items = []
        for p, lang in zip(prompts, langs):
            text = f"[START]{punc_norm(p)}[STOP]"
            it = {"prompt": f"[{lang.lower()}]{text}" if lang else text,
                  "multi_modal_data": {"conditionals": [cond_emb]}}
            if lang:
                it["tokenization_kwargs"] = {"language_id": lang.lower()}
            items.append(it)
        # the sampling clamp applies when any language id is given (mtl_tts.py:453-459); the waveforms get the alignment cap
        # and tail trim (mtl_tts.py:592-644) in render_waveforms
        # (per_prompt_clamp: each prompt clamped by its own language id, as separate calls would do: the streaming server)
        tokens_only = kwargs.pop("tokens_only", False)
        clamp = [bool(l) for l in langs] if kwargs.pop("per_prompt_clamp", False) else any(langs)
        toks = self._speech_tokens(items, clamp, temperature, top_p, max_tokens, repetition_penalty, *args, **kwargs)
        return toks if self.s3gen is None or tokens_only else self.render_waveforms(toks, s3gen_ref, diffusion_steps)
