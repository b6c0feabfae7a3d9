"""Model constants of the T3 speech-token decoder.

Every number here is a fact of the checkpoint format the reference serves, not a
tunable.  Sources in the reference tree:

* backbone hyper-parameters .......... t3-model/config.json:8-32 (hidden_size is
  written as 2048 there and reset to 1024 by src/chatterbox_vllm/models/t3/t3.py:310)
* vocabularies / special ids ......... src/chatterbox_vllm/models/t3/modules/t3_config.py:2-21
* prompt markers, cond rows, offset .. src/chatterbox_vllm/models/t3/t3.py:38-49
"""
from __future__ import annotations

from dataclasses import dataclass, field, asdict

# t3.py:38-49
PREFILL_COND_START_TOKEN = 695
PREFILL_COND_END_TOKEN = 696
PREFILL_END_TOKEN = 697
CONDITIONING_SIZE = 34
SPEECH_TOKEN_OFFSET = 1000

# t3_config.py:8-11
START_SPEECH_TOKEN = 6561
STOP_SPEECH_TOKEN = 6562
SPEECH_VOCAB = 8194
MAX_SPEECH_TOKENS = 4096
MAX_TEXT_TOKENS = 2048

TEXT_VOCAB_EN = 704
TEXT_VOCAB_MTL = 2454


@dataclass(frozen=True)
class T3ModelConfig:
    """Shape of one T3 checkpoint (english or multilingual)."""

    n_layers: int = 30
    d_model: int = 1024
    n_heads: int = 16
    head_dim: int = 64
    d_ff: int = 4096
    rms_eps: float = 1e-5
    rope_theta: float = 500000.0
    rope_factor: float = 8.0
    rope_low_freq_factor: float = 1.0
    rope_high_freq_factor: float = 4.0
    rope_original_max_pos: int = 8192
    text_vocab: int = TEXT_VOCAB_EN
    speech_vocab: int = SPEECH_VOCAB
    # learned position tables: t3.py:327-331 (max_text_tokens+2, max_speech_tokens+2+2)
    text_pos_rows: int = MAX_TEXT_TOKENS + 2
    speech_pos_rows: int = MAX_SPEECH_TOKENS + 4
    initializer_range: float = 0.02

    @property
    def is_multilingual(self) -> bool:
        return self.text_vocab == TEXT_VOCAB_MTL

    @classmethod
    def english(cls, **kw) -> "T3ModelConfig":
        return cls(text_vocab=TEXT_VOCAB_EN, **kw)

    @classmethod
    def multilingual(cls, **kw) -> "T3ModelConfig":
        return cls(text_vocab=TEXT_VOCAB_MTL, **kw)

    def to_dict(self) -> dict:
        return asdict(self)

    # ---- byte / flop accounting used by bench.py and the docs ----------------
    @property
    def layer_params(self) -> int:
        d, f = self.d_model, self.d_ff
        return 3 * d * d + d * d + 2 * d * f + f * d + 2 * d

    @property
    def backbone_params(self) -> int:
        return self.n_layers * self.layer_params + self.d_model

    @property
    def head_params(self) -> int:
        return self.speech_vocab * self.d_model

    @property
    def weight_bytes_per_step(self) -> int:
        """bf16 bytes every decode step has to stream (backbone + speech head)."""
        return 2 * (self.backbone_params + self.head_params)

    @property
    def kv_bytes_per_token(self) -> int:
        """bf16 K+V bytes one stream stores per position, all layers."""
        return 2 * self.n_heads * self.head_dim * 2 * self.n_layers
