"""OpenAI-style speech endpoint over the MI355X engine: the caller-side shell of the reference's
``openai_tts_server.py`` (route ``POST /v1/audio/speech`` :621-681, request schema :95-112, streaming generator
:438-528, chunker :144-327, PCM packing :117-141 / :330-339, language fallback :355-381, start-up / warm-up :533-616).

What is kept: the request fields and their defaults, the response media types and headers, the way a text is cut into
chunks (a short first chunk, larger ones after it, punctuation kept with the phrase before it, no chunk ending on a
function word), the 5 ms tail ramp per chunk, PCM16 little-endian frames of ``frame_ms``.

What differs, on purpose.  The reference synthesises a streamed request chunk after chunk, each through a T3 call of its
own and then S3Gen.  Here only the first chunk goes alone (time to first audio is its T3 call plus its S3Gen call, with
``first_chunk_diff_steps``); while its frames are being sent, ALL remaining chunks go through the T3 engine as one
batched call (continuous batching; every chunk keeps the ``max_tokens`` cap that a call of its own would have had) and
are then rendered in order.  With greedy sampling (``CHATTERBOX_DETERMINISTIC=1``) the bytes are those of chunk-by-chunk
calls, because the engine's ids do not depend on what else is in the batch.  Synthesis is serialised per engine by a
lock (the reference calls ``generate`` from worker threads without one, :501 / :661).

The text helpers and the byte packing are pure functions (no GPU, no web framework); ``create_app`` needs fastapi.
"""
from __future__ import annotations

import io
import os
import re
import threading
import wave
from typing import Any, Dict, Iterator, List, Optional

import torch

# ---- text chunking ---------------------------------------------------------------------------------------------------
_PHRASE_END = ".!?;:。！？；，、،؛؟…"                             # openai_tts_server.py:155
_PHRASE = re.compile("([^%s]*)([%s]?)" % (re.escape(_PHRASE_END), re.escape(_PHRASE_END)))
_BOUNDARY_PUNCT = ".,!?;:\\-—–…)\\]}"            # :233 / :290 (ASCII + dashes + ellipsis + closers)
_LEADING = re.compile("^[\\s%s。！？、，،؛؟]+" % _BOUNDARY_PUNCT)
_TRAILING = re.compile("[\\s%s]+$" % _BOUNDARY_PUNCT)
WEAK_END_WORDS = frozenset("a an the to of in on at and or but for nor so".split())     # :274-276


def _greedy_lines(pieces: List[str], limit: int, slice_oversize: bool) -> List[str]:
    """Join pieces with single spaces into lines of at most ``limit`` characters, in order.  A piece longer than the limit
    ends the current line and stands alone, or (``slice_oversize``) is cut into ``limit``-wide slices."""
    lines: List[str] = []
    cur = ""
    for piece in pieces:
        if len(piece) > limit:
            if cur:
                lines.append(cur)
                cur = ""
            if slice_oversize:
                lines.extend(piece[i:i + limit] for i in range(0, len(piece), limit))
            else:
                lines.append(piece)
        elif not cur:
            cur = piece
        elif len(cur) + 1 + len(piece) <= limit:
            cur = cur + " " + piece
        else:
            lines.append(cur)
            cur = piece
    if cur:
        lines.append(cur)
    return lines


def split_text_for_low_latency(text: str, max_chars: int = 120) -> List[str]:
    """Phrase-sized chunks of at most ``max_chars`` (openai_tts_server.py:144-217): phrases end at the punctuation of
    ``_PHRASE_END`` (a mark that follows nothing but whitespace is dropped, as there), are packed greedily, and what is
    still too long is re-packed word by word, words longer than the limit in fixed-width slices."""
    text = text.strip()
    if not text:
        return []
    phrases = []
    for body, mark in _PHRASE.findall(text):
        body = body.strip()
        if body:
            phrases.append(body + mark)
    packed = _greedy_lines(phrases, max_chars, slice_oversize=False) or [text]
    out: List[str] = []
    for line in packed:
        if len(line) <= max_chars:
            out.append(line)
        else:
            out.extend(_greedy_lines(line.split(), max_chars, slice_oversize=True))
    return [c for c in out if c]


def repair_chunk_boundaries(chunks: List[str]) -> List[str]:
    """No chunk starts with punctuation: a leading run of marks moves to the end of the chunk before it
    (openai_tts_server.py:220-259)."""
    out: List[str] = []
    for k, chunk in enumerate(str(c) for c in chunks):
        if k == 0:
            out.append(chunk.strip())
            continue
        lead = _LEADING.match(chunk)
        if lead:
            if lead.group(0).strip():
                out[-1] = (out[-1].rstrip() + lead.group(0)).rstrip()
            chunk = chunk[lead.end():]
        chunk = chunk.strip()
        if chunk:
            out.append(chunk)
    return [c for c in out if c]


def avoid_weak_endings(chunks: List[str]) -> List[str]:
    """No chunk but the last ends on a function word: such a word moves to the front of the next chunk, repeatedly, and a
    chunk emptied that way disappears, its punctuation going along (openai_tts_server.py:262-327)."""
    out = list(chunks)
    i = 0
    while i + 1 < len(out):
        here, after = out[i].strip(), out[i + 1].strip()
        stem = _TRAILING.sub("", here)
        words = stem.split()
        if not here or not after or not words or words[-1].lower() not in WEAK_END_WORDS:
            i += 1
            continue
        word, marks = words[-1], here[len(stem):]
        kept = stem[:stem.rfind(word)].rstrip()
        if kept:
            out[i] = (kept + marks).rstrip()
            out[i + 1] = (word + " " + after).strip()
        else:                                   # the whole chunk was that word: the boundary before it is looked at again
            out[i + 1] = (word + " " + (marks + " " + after).strip()).strip()
            del out[i]
    return [c for c in out if c and c.strip()]


def plan_chunks(text: str, first_chunk_chars: int = 30, chunk_chars: int = 120) -> List[str]:
    """The chunks of a streamed request (openai_tts_server.py:473-489): the first phrase cut at ``first_chunk_chars``, the
    rest of the text at ``chunk_chars``, then both boundary passes."""
    chunks: List[str] = []
    if first_chunk_chars and first_chunk_chars > 0:
        head = split_text_for_low_latency(text, max_chars=first_chunk_chars)
        if head:
            chunks.append(head[0])
            rest = text[len(head[0]):].strip()
            if rest:
                chunks.extend(split_text_for_low_latency(rest, max_chars=chunk_chars))
    else:
        chunks = split_text_for_low_latency(text, max_chars=chunk_chars)
    return avoid_weak_endings(repair_chunk_boundaries(chunks))


_SCRIPT_RANGES = (("he", ((0x0590, 0x05FF),)), ("ar", ((0x0600, 0x06FF),)), ("ru", ((0x0400, 0x04FF),)),
                  ("ko", ((0xAC00, 0xD7AF),)), ("ja", ((0x3040, 0x309F), (0x30A0, 0x30FF))), ("zh", ((0x4E00, 0x9FFF),)))


def detect_language(text: str) -> Optional[str]:
    """Script-based fallback when a request names no language (openai_tts_server.py:355-381): the first script in the
    order Hebrew, Arabic, Cyrillic, Hangul, kana, CJK ideographs that the text has a character of; None for anything else."""
    if not text:
        return None
    points = {ord(ch) for ch in text}
    for lang, ranges in _SCRIPT_RANGES:
        if any(lo <= p <= hi for p in points for lo, hi in ranges):
            return lang
    return None


# ---- audio packing ---------------------------------------------------------------------------------------------------
def float32_to_pcm16_bytes(wav: Any) -> bytes:
    """Waveform [1, N] or [N] in [-1, 1] -> little-endian 16-bit PCM, clamped, scaled by 32767 and truncated towards zero
    (openai_tts_server.py:117-127)."""
    x = torch.as_tensor(wav).detach().to("cpu", torch.float32).reshape(-1)
    return (x.clamp(-1.0, 1.0) * 32767.0).numpy().astype("<i2").tobytes()


def pack_wav_bytes(wav: Any, sr: int) -> bytes:
    """A mono 16-bit WAV file in memory (openai_tts_server.py:130-141)."""
    buf = io.BytesIO()
    with wave.open(buf, "wb") as f:
        f.setnchannels(1)
        f.setsampwidth(2)
        f.setframerate(sr)
        f.writeframes(float32_to_pcm16_bytes(wav))
    return buf.getvalue()


def frame_chunk_bytes(pcm: bytes, sr: int, frame_ms: int = 40) -> Iterator[bytes]:
    """PCM16 bytes in frames of ``frame_ms`` (the last one shorter) (openai_tts_server.py:330-339)."""
    step = 2 * int(sr * frame_ms / 1000)
    for at in range(0, len(pcm), step):
        yield pcm[at:at + step]


def tail_ramp(wav: torch.Tensor, sr: int, fade_ms: int = 5) -> torch.Tensor:
    """The last ``fade_ms`` of a chunk scaled by a ramp from 1.0 to 0.95 (openai_tts_server.py:511-518)."""
    n = int(sr * fade_ms / 1000)
    if n <= 0 or wav.numel() <= n:
        return wav
    out = wav.clone()
    out[:, -n:] *= torch.linspace(1.0, 0.95, steps=n, dtype=wav.dtype, device=wav.device).unsqueeze(0)
    return out


# ---- synthesis service -----------------------------------------------------------------------------------------------
class SpeechService:
    """One engine (``ChatterboxTTS`` or ``ChatterboxMultilingualTTS`` with a token -> waveform chain attached) behind a
    lock, with the two ways the endpoint uses it: a whole text in one call, or a text streamed chunk by chunk."""

    def __init__(self, tts: Any):
        if getattr(tts, "s3gen", None) is None:
            raise ValueError("the speech endpoint needs an engine with the token -> waveform chain attached (s3gen)")
        self.tts = tts
        self._lock = threading.Lock()

    @property
    def sr(self) -> int:
        return self.tts.sr

    def _multilingual(self) -> bool:
        return getattr(self.tts, "variant", "english") == "multilingual"

    def _language(self, text: str, language_id: Optional[str]) -> Optional[str]:
        if self._multilingual() and not (language_id and str(language_id).strip()):
            return detect_language(text) or language_id             # openai_tts_server.py:406-411
        return language_id

    def synthesize(self, text: str, *, language_id: Optional[str] = "en", temperature: float = 0.8,
                   exaggeration: float = 0.5, diffusion_steps: int = 10, audio_prompt_path: Optional[str] = None) -> torch.Tensor:
        """One text -> one waveform [1, N] (openai_tts_server.py:392-435: a single ``generate`` call with
        ``max_tokens = max_model_len``)."""
        with self._lock:
            waves = self.tts.generate(prompts=text, audio_prompt_path=audio_prompt_path,
                                      language_id=self._language(text, language_id), temperature=temperature,
                                      exaggeration=exaggeration, diffusion_steps=diffusion_steps,
                                      max_tokens=self.tts.max_model_len)
        if not waves:
            raise RuntimeError("No audio generated")
        return waves[0]

    def stream_waveforms(self, text: str, *, language_id: Optional[str] = "en", temperature: float = 0.8,
                         exaggeration: float = 0.5, diffusion_steps: int = 10, audio_prompt_path: Optional[str] = None,
                         first_chunk_diff_steps: Optional[int] = None, first_chunk_chars: int = 30,
                         chunk_chars: int = 120) -> Iterator[torch.Tensor]:
        """The chunks of ``plan_chunks`` as waveforms, in order, each with the tail ramp.  The first chunk is synthesised
        alone; the speech tokens of all the others come from one batched engine call, issued when the consumer asks for
        the second waveform, and each is rendered when its turn comes."""
        chunks = plan_chunks(text, first_chunk_chars, chunk_chars)
        if not chunks:
            return
        tts = self.tts
        # the reference synthesises chunk after chunk through _synthesize_one (openai_tts_server.py:501-509), so its
        # language fallback looks at each chunk's own text (:406-411): mixed-script input gets a language per chunk
        langs = [self._language(c, language_id) for c in chunks]
        first_steps = diffusion_steps if first_chunk_diff_steps is None else max(1, int(first_chunk_diff_steps))
        yield tail_ramp(self.synthesize(chunks[0], language_id=langs[0], temperature=temperature, exaggeration=exaggeration,
                                        diffusion_steps=first_steps, audio_prompt_path=audio_prompt_path), self.sr)
        rest = chunks[1:]
        if not rest:
            return
        with self._lock:
            ref, cond = tts.get_audio_conditionals(audio_prompt_path)
            per_chunk = dict(language_id=langs[1:], per_prompt_clamp=True) if self._multilingual() else dict(language_id=language_id)
            toks = tts.generate_with_conds(prompts=rest, s3gen_ref=ref, cond_emb=cond, **per_chunk,
                                           temperature=temperature, exaggeration=exaggeration,
                                           max_tokens=tts.max_model_len, **self._facade_defaults(),
                                           per_prompt_max_tokens=True, tokens_only=True)   # rendering follows per chunk
        for tok in toks:
            with self._lock:
                wav = tts.render_waveforms([tok], ref, diffusion_steps)[0]
            yield tail_ramp(wav, self.sr)

    def _facade_defaults(self) -> Dict[str, Any]:
        """``generate``'s own default for ``top_p`` (tts.py:259-288: 0.8 where ``generate_with_conds`` has 1.0), so that a
        chunk taken through the batched call is sampled as ``generate`` would sample it."""
        import inspect
        top_p = inspect.signature(type(self.tts).generate).parameters.get("top_p")
        return {} if top_p is None or top_p.default is inspect.Parameter.empty else {"top_p": top_p.default}

    def stream_pcm_frames(self, text: str, *, frame_ms: int = 20, primer_silence_ms: int = 0, **synthesis) -> Iterator[bytes]:
        """PCM16 frames of a streamed request (openai_tts_server.py:438-528): optional leading silence so that headers and a
        first body chunk leave at once, then every chunk's waveform in frames of ``frame_ms``."""
        if primer_silence_ms and primer_silence_ms > 0:
            n = int(self.sr * primer_silence_ms / 1000)
            if n > 0:
                yield bytes(2 * n)
        for wav in self.stream_waveforms(text, **synthesis):
            for frame in frame_chunk_bytes(float32_to_pcm16_bytes(wav), self.sr, frame_ms):
                if frame:
                    yield frame

    def warm_up(self, text: Optional[str] = None, diffusion_steps: Optional[int] = None) -> None:
        """Default voice + one tiny synthesis, as the reference does at start-up (openai_tts_server.py:582-603)."""
        text = text or os.environ.get("CHATTERBOX_WARMUP_TEXT", "Hello")
        steps = diffusion_steps or int(os.environ.get("CHATTERBOX_WARMUP_DIFF_STEPS", "2"))
        with self._lock:
            self.tts.get_audio_conditionals(None)
            self.tts.generate(prompts=text, language_id="en", diffusion_steps=steps,
                              max_tokens=min(32, self.tts.max_model_len))

    def shutdown(self) -> None:
        with self._lock:
            self.tts.shutdown()


def load_engine_from_env() -> Any:
    """The engine the reference's start-up hook loads (openai_tts_server.py:533-580): CHATTERBOX_VARIANT
    ("multilingual" | "english"), CHATTERBOX_USE_LOCAL_CKPT (a checkpoint directory; else the hub download),
    CHATTERBOX_DEVICE, CHATTERBOX_S3GEN_FP16, CHATTERBOX_COMPILE."""
    from .tts import ChatterboxTTS
    from .mtl_tts import ChatterboxMultilingualTTS
    on = lambda name: os.environ.get(name, "0").strip().lower() in ("1", "true", "yes", "on")
    variant = os.environ.get("CHATTERBOX_VARIANT", "multilingual").strip().lower()
    kw = dict(target_device=os.environ.get("CHATTERBOX_DEVICE", "cuda"), s3gen_use_fp16=on("CHATTERBOX_S3GEN_FP16"),
              compile=on("CHATTERBOX_COMPILE"))
    local = os.environ.get("CHATTERBOX_USE_LOCAL_CKPT", "").strip() or None
    if variant == "multilingual":
        return (ChatterboxMultilingualTTS.from_local(ckpt_dir=local, **kw) if local
                else ChatterboxMultilingualTTS.from_pretrained(**kw))
    return (ChatterboxTTS.from_local(ckpt_dir=local, variant="english", **kw) if local
            else ChatterboxTTS.from_pretrained(**kw))


# ---- HTTP shell ------------------------------------------------------------------------------------------------------
try:
    from pydantic import BaseModel, Field
except ImportError:                                                  # the helpers above work without the web stack
    BaseModel = None

if BaseModel is not None:
    class SpeechRequest(BaseModel):                                 # openai_tts_server.py:95-112
        model: Optional[str] = Field(default="gpt-4o-mini-tts")      # ignored, as `voice` is: OpenAI compatibility
        voice: Optional[str] = Field(default="alloy")
        input: str
        format: Optional[str] = Field(default="wav", pattern="^(wav|pcm16)$")
        response_format: Optional[str] = Field(default=None, pattern="^(wav|pcm16|pcm)$")
        stream: Optional[bool] = Field(default=False)
        language_id: Optional[str] = Field(default="en")
        temperature: Optional[float] = Field(default=0.8)
        exaggeration: Optional[float] = Field(default=0.5)
        diffusion_steps: Optional[int] = Field(default=10)
        audio_prompt_path: Optional[str] = Field(default=None)
        watermark: Optional[str] = Field(default="off", pattern="^(on|off)$")     # no watermark is applied either way
        primer_silence_ms: Optional[int] = Field(default=0, ge=0, le=200)
        first_chunk_diff_steps: Optional[int] = Field(default=10, ge=1)
        first_chunk_chars: Optional[int] = Field(default=30, ge=10, le=200)
        chunk_chars: Optional[int] = Field(default=120, ge=40, le=400)
        frame_ms: Optional[int] = Field(default=20, ge=10, le=60)


def create_app(service: Optional[SpeechService] = None, warm_up: bool = True):
    """The FastAPI application.  With ``service`` None the engine is loaded from the environment at start-up and shut down
    with the application, as in the reference; a given service is used as it is (and left running at shutdown)."""
    import asyncio
    from contextlib import asynccontextmanager
    from fastapi import FastAPI, HTTPException
    from fastapi.responses import Response, StreamingResponse

    state: Dict[str, Any] = {"service": service}

    @asynccontextmanager
    async def lifespan(app):
        owned = state["service"] is None
        if owned:
            state["service"] = SpeechService(load_engine_from_env())
        if warm_up:
            try:
                await asyncio.to_thread(state["service"].warm_up)
            except Exception as e:                                   # a failed warm-up does not stop the server (:602-603)
                print(f"[WARN] Warmup failed: {e}")
        yield
        if owned and state["service"] is not None:
            state["service"].shutdown()
            state["service"] = None

    app = FastAPI(title="OpenAI-compatible TTS on the MI355X engine", lifespan=lifespan)

    def synthesis_args(req) -> Dict[str, Any]:
        return dict(language_id=req.language_id, temperature=req.temperature or 0.8, exaggeration=req.exaggeration or 0.5,
                    diffusion_steps=req.diffusion_steps or 10, audio_prompt_path=req.audio_prompt_path)

    @app.post("/v1/audio/speech")
    async def create_speech(req: SpeechRequest):
        svc: Optional[SpeechService] = state["service"]
        if svc is None:
            raise HTTPException(status_code=503, detail="TTS engine not initialized")
        fmt = req.response_format or req.format or "wav"
        fmt = "pcm16" if fmt == "pcm" else fmt
        pcm_type = f"audio/pcm;rate={svc.sr};channels=1"
        if req.stream:                                               # linear PCM whatever format was asked for (:633-657)
            frames = svc.stream_pcm_frames(
                req.input, frame_ms=req.frame_ms or 20, primer_silence_ms=req.primer_silence_ms or 0,
                first_chunk_diff_steps=req.first_chunk_diff_steps if req.first_chunk_diff_steps is not None else 10,
                first_chunk_chars=req.first_chunk_chars or 30, chunk_chars=req.chunk_chars or 120, **synthesis_args(req))

            async def body():
                # the generator blocks on the GPU between chunks: each step runs on a worker thread
                done = object()
                while True:
                    frame = await asyncio.to_thread(next, frames, done)
                    if frame is done:
                        return
                    yield frame
            return StreamingResponse(body(), media_type=pcm_type, headers={
                "X-Accel-Buffering": "no", "Cache-Control": "no-cache", "Pragma": "no-cache"})
        try:
            wav = await asyncio.to_thread(lambda: svc.synthesize(req.input, **synthesis_args(req)))
        except ValueError as e:
            raise HTTPException(status_code=400, detail=str(e))
        except Exception as e:
            raise HTTPException(status_code=500, detail=f"Synthesis error: {e}")
        if fmt == "pcm16":
            return Response(content=float32_to_pcm16_bytes(wav), media_type=pcm_type)
        return Response(content=pack_wav_bytes(wav, svc.sr), media_type="audio/wav")

    return app


def main(argv: Optional[List[str]] = None) -> None:
    """``python -m chatterbox_vllm_amd.server [--host H] [--port P]`` (openai_tts_server.py:697-729: port 8000)."""
    import argparse
    import uvicorn
    ap = argparse.ArgumentParser(description=main.__doc__)
    ap.add_argument("--host", default="0.0.0.0")
    ap.add_argument("--port", type=int, default=8000)
    args = ap.parse_args(argv)
    uvicorn.run(create_app(), host=args.host, port=args.port, log_level="info")


if __name__ == "__main__":
    main()
