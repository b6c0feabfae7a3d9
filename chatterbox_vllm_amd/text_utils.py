"""Prompt text normalisation and the language table.

Behaviour follows src/chatterbox_vllm/text_utils.py:2-41 (``punc_norm``) and :44-68
(``SUPPORTED_LANGUAGES``) of the reference; tests/test_host_cpu.py pins it against strings
captured from the reference function (tests/golden/ref_text_fixture.json).
"""
from __future__ import annotations

import math
import os
import re

EMPTY_PROMPT_TEXT = "You need to add some text for me to talk."

# (needle, replacement), applied in this order: later rules see earlier results
_PUNCTUATION_RULES = (
    ("...", ", "), ("…", ", "), (":", ","), (" - ", ", "), (";", ", "),
    ("—", "-"), ("–", "-"), (" ,", ","),
    ("“", '"'), ("”", '"'), ("‘", "'"), ("’", "'"),
)
_SENTENCE_ENDERS = (".", "!", "?", "-", ",", "、", "，", "。", "？", "！")

SUPPORTED_LANGUAGES = {
    "ar": "Arabic", "da": "Danish", "de": "German", "el": "Greek", "en": "English", "es": "Spanish",
    "fi": "Finnish", "fr": "French", "he": "Hebrew", "hi": "Hindi", "it": "Italian", "ja": "Japanese",
    "ko": "Korean", "ms": "Malay", "nl": "Dutch", "no": "Norwegian", "pl": "Polish", "pt": "Portuguese",
    "ru": "Russian", "sv": "Swedish", "sw": "Swahili", "tr": "Turkish", "zh": "Chinese",
}


def punc_norm(text: str) -> str:
    """Capitalise, squeeze whitespace, map rare punctuation, make sure the text ends a sentence."""
    if not text:
        return EMPTY_PROMPT_TEXT
    if text[0].islower():
        text = text[0].upper() + text[1:]
    text = " ".join(text.split())
    for needle, repl in _PUNCTUATION_RULES:
        text = text.replace(needle, repl)
    text = text.rstrip(" ")
    if not text.endswith(_SENTENCE_ENDERS):
        text += "."
    return text


def _env_flag(name: str, default: str = "0") -> bool:
    return os.environ.get(name, default).lower() in ("1", "true", "yes", "on")


def nonspace_chars(prompt: str) -> int:
    """Characters that count towards the length estimate: bracketed tags removed, whitespace
    ignored (src/chatterbox_vllm/tts.py:381-382)."""
    return sum(1 for c in re.sub(r"\[[^\]]+\]", "", prompt) if not c.isspace())


def guard_cap(prompt: str) -> int:
    """Per-prompt speech-token ceiling ceil(clamp(ceil(chars * 2.2), 64, 1200) * 1.6), with the
    reference's environment overrides (tts.py:373-386)."""
    tpc = float(os.environ.get("CHATTERBOX_TOKENS_PER_CHAR", "2.2"))
    tmin = int(os.environ.get("CHATTERBOX_TOKENS_MIN", "64"))
    tmax = int(os.environ.get("CHATTERBOX_TOKENS_MAX", "1200"))
    mult = float(os.environ.get("CHATTERBOX_TOKENS_GUARD_MULT", "1.6"))
    est = max(tmin, min(tmax, int(math.ceil(nonspace_chars(prompt) * tpc))))
    return int(math.ceil(est * mult))


def batch_max_tokens(prompts, max_tokens: int, max_model_len: int) -> int:
    """One ``max_tokens`` for the whole call: the smallest per-prompt guard plus a margin,
    capped by the caller's and the model's limits (tts.py:378-393)."""
    margin = int(os.environ.get("CHATTERBOX_PRE_GUARD_MARGIN", "16"))
    ceiling = min(max_tokens, max_model_len)
    caps = [guard_cap(p) for p in prompts]
    if not caps:
        return ceiling
    return min(max(1, min(caps) + max(0, margin)), ceiling)
