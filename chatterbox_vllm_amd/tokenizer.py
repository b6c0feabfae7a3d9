"""Text tokenizers feeding the T3 prompt: str -> text ids.

Both wrap a ``tokenizers`` BPE vocabulary file that ships with the checkpoint (it is not
part of this package).  Behaviour follows the reference's vLLM-facing tokenizers:

* English ........ src/chatterbox_vllm/models/t3/entokenizer.py:93-95 (" " -> "[SPACE]", BPE)
  and its vocabulary lookup order :47-83 (given directory, then ./t3-model).
* Multilingual ... src/chatterbox_vllm/models/t3/mtl_tokenizer.py:307-390 (lower-case + NFKD
  with [START]/[STOP] restored, leading "[xx]" language tag, per-language preprocessing,
  " " -> "[SPACE]") and :256-294 for the vocabulary lookup order.

Language preprocessors that need assets or packages outside this repository (Cangjie table +
pkuseg for zh, pykakasi for ja, dicta for he, a stresser for ru) are used when importable /
present and skipped with a warning otherwise, as in the reference (:81-83, :97-99, :164-166,
:215-217).  Korean Jamo decomposition is arithmetic and always on (:105-123).
"""
from __future__ import annotations

import json
import logging
import os
import re
from typing import List, Optional
from unicodedata import category, normalize

from .text_utils import SUPPORTED_LANGUAGES

SOT, EOT, UNK, SPACE = "[START]", "[STOP]", "[UNK]", "[SPACE]"
log = logging.getLogger(__name__)


def _first_existing(paths):
    for p in paths:
        if os.path.isfile(p):
            return p
    return None


class EnTokenizer:
    VOCAB_FILES = ("tokenizer.json",)
    CWD_MODEL_DIR = "t3-model"

    def __init__(self, vocab_file: str):
        from tokenizers import Tokenizer
        self.tokenizer = Tokenizer.from_file(vocab_file)
        voc = self.tokenizer.get_vocab()
        assert SOT in voc and EOT in voc, "vocabulary lacks [START]/[STOP]"

    @classmethod
    def from_pretrained(cls, path: Optional[str] = None, **kw):
        cands = []
        if path and os.path.isdir(path):
            cands += [os.path.join(path, f) for f in cls.VOCAB_FILES]
        elif path and os.path.isfile(path):
            cands.append(path)
        cwd_dir = os.path.join(os.getcwd(), cls.CWD_MODEL_DIR)
        cands += [os.path.join(cwd_dir, f) for f in cls.VOCAB_FILES]
        vocab = _first_existing(cands)
        if vocab is None:
            raise FileNotFoundError(f"{cls.__name__}: could not locate {' or '.join(cls.VOCAB_FILES)}")
        return cls(vocab, **kw)

    def prepare(self, text: str, language_id: Optional[str] = None) -> str:
        return text.replace(" ", SPACE)

    def encode(self, text: str, language_id: Optional[str] = None, **_) -> List[int]:
        return self.tokenizer.encode(self.prepare(text, language_id)).ids

    __call__ = encode

    def decode(self, ids) -> str:
        txt = self.tokenizer.decode(list(ids), skip_special_tokens=False)
        return txt.replace(" ", "").replace(SPACE, " ").replace(EOT, "").replace(UNK, "")

    @property
    def vocab_size(self) -> int:
        return self.tokenizer.get_vocab_size()

    @property
    def max_token_id(self) -> int:
        return max(self.tokenizer.get_vocab().values())


def korean_to_jamo(text: str) -> str:
    """Hangul syllable -> initial/medial/final Jamo (U+AC00 block arithmetic)."""
    out = []
    for ch in text:
        o = ord(ch)
        if 0xAC00 <= o <= 0xD7AF:
            base = o - 0xAC00
            out.append(chr(0x1100 + base // 588))
            out.append(chr(0x1161 + (base % 588) // 28))
            if base % 28:
                out.append(chr(0x11A7 + base % 28))
        else:
            out.append(ch)
    return "".join(out).strip()


class _Cangjie:
    """zh: Han characters -> "[cj_x]...[cj_.]" code tokens from a Cangjie5_TC.json table."""

    def __init__(self, model_dir: Optional[str]):
        self.word2cj, self.cj2word, self.segmenter = {}, {}, None
        path = _first_existing([os.path.join(model_dir, "Cangjie5_TC.json")] if model_dir else [])
        if path is None:
            log.warning("Cangjie5_TC.json not found - Chinese text is tokenized without Cangjie codes")
        else:
            with open(path, encoding="utf-8") as fp:
                for entry in json.load(fp):
                    word, code = entry.split("\t")[:2]
                    self.word2cj[word] = code
                    self.cj2word.setdefault(code, []).append(word)
        try:
            from spacy_pkuseg import pkuseg
            self.segmenter = pkuseg()
        except ImportError:
            log.warning("pkuseg not available - Chinese segmentation skipped")

    def __call__(self, text: str) -> str:
        if self.segmenter is not None:
            text = " ".join(self.segmenter.cut(text))
        out = []
        for ch in text:
            code = self.word2cj.get(ch) if category(ch) == "Lo" else None
            if code is None:
                out.append(ch)
                continue
            idx = self.cj2word[code].index(ch)
            code = code + (str(idx) if idx > 0 else "")
            out.append("".join(f"[cj_{c}]" for c in code) + "[cj_.]")
        return "".join(out)


def _optional(module: str, what: str):
    try:
        return __import__(module, fromlist=["_"])
    except ImportError:
        log.warning("%s not available - %s skipped", module, what)
        return None


class MTLTokenizer(EnTokenizer):
    VOCAB_FILES = ("grapheme_mtl_merged_expanded_v1.json", "mtl_tokenizer.json")
    CWD_MODEL_DIR = "t3-multilingual-model"

    def __init__(self, vocab_file: str):
        super().__init__(vocab_file)
        self._model_dir = os.path.dirname(os.path.abspath(vocab_file))
        self._cangjie = None
        self._kakasi = None

    @staticmethod
    def split_language_tag(text: str, language_id: Optional[str]):
        """A leading "[xx]" is a language tag only when xx is a supported code (mtl_tokenizer.py:370-374)."""
        if not language_id and text.startswith("[") and "]" in text[:8]:
            tag = text[1:text.index("]")].lower()
            if tag in SUPPORTED_LANGUAGES:
                return text[text.index("]") + 1:], tag
        return text, language_id

    @staticmethod
    def preprocess(text: str) -> str:
        text = normalize("NFKD", text.lower())
        return re.sub(r"\[stop\]", EOT, re.sub(r"\[start\]", SOT, text))

    def _language_rules(self, text: str, lang: Optional[str]) -> str:
        if lang == "zh":
            if self._cangjie is None:
                self._cangjie = _Cangjie(self._model_dir)
            text = self._cangjie(text)
        elif lang == "ko":
            text = korean_to_jamo(text)
        elif lang == "ja":
            mod = _optional("pykakasi", "Japanese kana conversion")
            if mod is not None:
                self._kakasi = self._kakasi or mod.kakasi()
                parts = []
                for r in self._kakasi.convert(text):
                    orig, hira = r["orig"], r["hira"]
                    if any(0x4E00 <= ord(c) <= 0x9FFF for c in orig):
                        parts.append((" " + hira) if hira and hira[0] in "はへ" else hira)
                    else:
                        parts.append(orig)
                text = normalize("NFKD", "".join(parts))
        elif lang == "he":
            mod = _optional("dicta_onnx", "Hebrew diacritics")
            if mod is not None:
                text = mod.Dicta().add_diacritics(text)
        elif lang == "ru":
            mod = _optional("russian_text_stresser.text_stresser", "Russian stress marks")
            if mod is not None:
                text = mod.RussianTextStresser().stress_text(text)
        if lang:
            tag = f"[{lang.lower()}]"
            if not text.startswith(tag):
                text = tag + text
        return text

    def prepare(self, text: str, language_id: Optional[str] = None) -> str:
        text, lang = self.split_language_tag(text, language_id)
        return self._language_rules(self.preprocess(text), lang).replace(" ", SPACE)
