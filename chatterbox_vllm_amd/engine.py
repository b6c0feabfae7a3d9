"""``T3Engine``: the object ``ChatterboxTTS`` holds where the reference holds ``vllm.LLM``.

It accepts the same call the reference makes
(src/chatterbox_vllm/tts.py:365-370, 408-418)::

    outs = engine.generate(
        [{"prompt": "[START]Hello.[STOP]", "multi_modal_data": {"conditionals": [cond_emb]}}, ...],
        sampling_params=SamplingParams(temperature=..., stop_token_ids=[7562], max_tokens=..., top_p=...,
                                       repetition_penalty=..., min_p=..., seed=...))
    outs[i].outputs[0].token_ids      # ids in the +1000 space, stop id included

and returns results in input order.  All arithmetic runs in ``libt3hip.so`` (HIP kernels for
gfx950 behind a C ABI); this module only tokenises, queues and collects.
"""
from __future__ import annotations

import ctypes as C
import os
import threading
from dataclasses import dataclass, field
from typing import Callable, Dict, Iterable, List, Optional, Sequence, Union

import torch

from . import _capi
from .config import (T3ModelConfig, CONDITIONING_SIZE, SPEECH_TOKEN_OFFSET, STOP_SPEECH_TOKEN)
from .weights import rope_table, route_checkpoint


@dataclass
class SamplingParams:
    """The fields of ``vllm.SamplingParams`` the T3 path uses, with vLLM's defaults."""
    temperature: float = 1.0
    top_p: float = 1.0
    min_p: float = 0.0
    repetition_penalty: float = 1.0
    max_tokens: int = 16
    stop_token_ids: Optional[Sequence[int]] = None
    seed: Optional[int] = None
    # Accepted for callers that pass them (the reference's server does for the multilingual model,
    # openai_tts_server.py:415-421) and without effect here: they steer vLLM's detokenizer and its tokenizer-level
    # end-of-sequence id, neither of which the speech-token path has (it stops on stop_token_ids).
    ignore_eos: bool = False
    skip_special_tokens: bool = True
    spaces_between_special_tokens: bool = True

    def __post_init__(self):
        if self.temperature < 0:
            raise ValueError("temperature must be non-negative")
        if not 0 < self.top_p <= 1:
            raise ValueError("top_p must be in (0, 1]")
        if not 0 <= self.min_p <= 1:
            raise ValueError("min_p must be in [0, 1]")
        if self.repetition_penalty <= 0:
            raise ValueError("repetition_penalty must be positive")
        if self.max_tokens < 1:
            raise ValueError("max_tokens must be at least 1")


@dataclass
class CompletionOutput:
    index: int
    token_ids: List[int]
    finish_reason: Optional[str] = None      # "stop" | "length"


@dataclass
class RequestOutput:
    request_id: str
    prompt: Optional[str]
    prompt_token_ids: List[int]
    outputs: List[CompletionOutput] = field(default_factory=list)
    finished: bool = True


_FINISH = {1: "stop", 2: "length", 3: "error"}


SCHEDULING_POLICIES = ("fcfs", "longest_first", "longest_prompt_first")     # index = the C ABI's policy number


class T3Engine:
    """One engine per GPU.  ``generate`` calls from several threads (the reference's server calls it from
    worker threads without a lock, openai_tts_server.py:501, 661) are serialised here: one call runs its
    requests dry before the next starts."""

    def __init__(self, model_config: T3ModelConfig, state_dict: Dict[str, torch.Tensor],
                 tokenizer: Optional[Callable[..., List[int]]] = None, *,
                 max_model_len: int = 1000, max_num_seqs: int = 32, cfg_scale: Optional[float] = None,
                 cfg_paired: Optional[bool] = None, device: int = 0, enforce_eager: bool = False,
                 kv_pages: int = 0, max_prefill_rows: int = 2048, quantization: Optional[str] = None,
                 scheduling_policy: str = "fcfs", uncond_attends: Optional[bool] = None, prefix_sharing: bool = True,
                 max_voices: int = 0, stream: Optional[int] = None, max_num_batched_tokens: Optional[int] = None,
                 **_ignored_vllm_kwargs):
        if cfg_scale is None:                      # models/t3/t3.py:343
            cfg_scale = float(os.environ.get("CHATTERBOX_CFG_SCALE", "0.5"))
        if cfg_paired is None:
            cfg_paired = True                      # the reference's model doubles every row into a cond and an uncond half (t3.py:762-772)
        if uncond_attends is None:
            # What the uncond half attends to is decided inside the serving stack the reference runs on, which is not part
            # of it: the documented intent is a second stream with its own K/V (the default here); the other reading, an
            # uncond row without attention output or K/V, is selected with CHATTERBOX_UNCOND_ATTENDS=0 (or the argument).
            uncond_attends = os.environ.get("CHATTERBOX_UNCOND_ATTENDS", "1") not in ("0", "false", "False")
        self.model_config = model_config
        self.tokenizer = tokenizer
        self.max_model_len = int(max_model_len)
        self.max_num_seqs = int(max_num_seqs)
        self.cfg_scale = float(cfg_scale)
        self.cfg_paired = bool(cfg_paired)
        # uncond_attends=False: the uncond half is a context-free row (zero attention output at every layer, no KV),
        # the arithmetic a serving stack performs when it attends only the rows its scheduler counted
        self.uncond_attends = bool(uncond_attends)
        # prefix_sharing: the K/V of the 34 conditioning positions are computed once per distinct cond_emb and shared by
        # both streams of a request and by all requests of that voice; ids and logits are bit-identical either way
        self.prefix_sharing = bool(prefix_sharing)
        self.device = int(device)
        if quantization not in (None, "int8"):
            raise ValueError(f"unsupported quantization {quantization!r}: None or 'int8' (W8A16, per output channel)")
        self.quantization = quantization
        self._lib = _capi.load()
        m = model_config
        cfg = _capi.T3Config(
            n_layers=m.n_layers, d_model=m.d_model, n_heads=m.n_heads, head_dim=m.head_dim, d_ff=m.d_ff,
            text_vocab=m.text_vocab, speech_vocab=m.speech_vocab, text_pos_rows=m.text_pos_rows,
            speech_pos_rows=m.speech_pos_rows, max_model_len=self.max_model_len, max_num_seqs=self.max_num_seqs,
            cfg_paired=(1 if self.uncond_attends else 2) if self.cfg_paired else 0, max_prefill_rows=int(max_prefill_rows), device=self.device,
            use_graphs=int(not enforce_eager), weight_int8=int(quantization == "int8"), cfg_scale=self.cfg_scale, rms_eps=m.rms_eps,
            kv_pages=int(kv_pages), prefix_sharing=int(self.prefix_sharing), max_voices=int(max_voices))
        self._h = C.c_void_p()
        # stream: a raw hipStream_t (e.g. torch.cuda.Stream().cuda_stream) the engine enqueues its decode iterations
        # on, in order with the caller's work there; None = a stream of the engine's own
        if stream is None:
            _capi.check(self._lib, self._lib.t3_engine_create(C.byref(cfg), C.byref(self._h)))
        else:
            _capi.check(self._lib, self._lib.t3_engine_create_on_stream(C.byref(cfg), C.c_void_p(int(stream)), C.byref(self._h)))
        self.set_schedule_policy(scheduling_policy)
        self.max_num_batched_tokens = None
        self._next_id = 0
        self._generate_lock = threading.Lock()
        self._load(state_dict)
        if max_num_batched_tokens:
            self.set_max_num_batched_tokens(max_num_batched_tokens)

    # ---- weights -----------------------------------------------------------------
    def _load_tensor(self, name: str, t: torch.Tensor) -> None:
        # a tensor already on the engine's GPU is handed over as a device pointer (no trip through host memory)
        on_device = t.is_cuda and t.device.index == self.device
        t = t.detach().to(dtype=torch.bfloat16).contiguous() if on_device else \
            t.detach().to(device="cpu", dtype=torch.bfloat16).contiguous()
        shape = (C.c_int64 * t.dim())(*t.shape)
        if on_device:
            torch.cuda.current_stream(self.device).synchronize()      # the library copies on its own: producers must be done
            load = self._lib.t3_engine_load_tensor_dev
        else:
            load = self._lib.t3_engine_load_tensor
        _capi.check(self._lib, load(self._h, name.encode(), _capi.ptr(t), shape, t.dim(), 0))

    def _load(self, state_dict: Dict[str, torch.Tensor]) -> None:
        keep, self.skipped_tensors = route_checkpoint(state_dict.items(), self.model_config)
        for name, t in keep.items():
            self._load_tensor(name, t)
        self._load_tensor("rope.cos_sin", rope_table(self.model_config, self.max_model_len))
        _capi.check(self._lib, self._lib.t3_engine_finalize(self._h))

    # ---- request level -------------------------------------------------------------
    def add_request(self, text_ids: Sequence[int], cond_emb: torch.Tensor, sp: SamplingParams,
                    request_id: Optional[int] = None) -> int:
        assert cond_emb is not None, "Conditionals are required for prefill"            # t3.py:248
        assert cond_emb.shape[0] == CONDITIONING_SIZE, "Conditionals must be CONDITIONING_SIZE tokens long"  # t3.py:250
        assert cond_emb.shape[1] == self.model_config.d_model
        rid = self._next_id if request_id is None else int(request_id)
        self._next_id = max(self._next_id, rid) + 1
        ids = (C.c_int32 * len(text_ids))(*[int(t) for t in text_ids])
        cond = cond_emb.detach().to(device="cpu", dtype=torch.float32).contiguous()
        stop = list(sp.stop_token_ids or [])
        if len(stop) > 5:
            raise ValueError("the T3 engine supports at most 5 stop token ids per request")
        s = _capi.T3Sampling(temperature=sp.temperature, top_p=sp.top_p, min_p=sp.min_p,
                             repetition_penalty=sp.repetition_penalty, max_tokens=int(sp.max_tokens),
                             stop_token=int(stop[0]) if stop else -1, has_seed=int(sp.seed is not None),
                             n_extra_stop=max(0, len(stop) - 1), extra_stop_tokens=(C.c_int32 * 4)(*([int(t) for t in stop[1:]] + [0] * (5 - max(1, len(stop))))),
                             seed=int(sp.seed or 0) & (2 ** 64 - 1))
        _capi.check(self._lib, self._lib.t3_engine_add_request(
            self._h, rid, ids, len(text_ids), C.cast(_capi.ptr(cond), C.POINTER(C.c_float)), C.byref(s)))
        return rid

    def step(self):
        """One scheduler iteration -> list of (request_id, token_id, finish_reason | None).

        Steps are pipelined one deep inside the library: a call launches the next step and reports the
        previous one, so the first call after admission may report nothing and a call that switches to a
        synchronous mode (timing, logit taps) may report two steps' tokens, in order.

        A request whose logits have no finite maximum (non-finite weights or conditioning) is ended by the library and
        reported once with finish reason "error" (its token is meaningless); the other requests of the step are
        reported as usual, so a serving loop loses nothing of theirs.  ``generate`` raises for such a request after
        the rest of its call has finished."""
        cap = 2 * self.max_num_seqs
        if getattr(self, "_step_bufs", None) is None:
            self._step_bufs = ((C.c_int64 * cap)(), (C.c_int32 * cap)(), (C.c_int32 * cap)(), C.c_int32(0))
        rid, tok, fin, n = self._step_bufs
        _capi.check(self._lib, self._lib.t3_engine_step(self._h, rid, tok, fin, cap, C.byref(n)))
        return [(rid[i], tok[i], _FINISH.get(fin[i])) for i in range(n.value)]

    def has_unfinished(self) -> bool:
        return bool(self._lib.t3_engine_has_unfinished(self._h))

    def abort(self, request_id: int) -> None:
        _capi.check(self._lib, self._lib.t3_engine_abort(self._h, int(request_id)))

    def stats(self) -> _capi.T3StepStats:
        s = _capi.T3StepStats()
        _capi.check(self._lib, self._lib.t3_engine_get_stats(self._h, C.byref(s)))
        return s

    def set_schedule_policy(self, policy: str) -> None:
        """Admission order for requests added from now on: 'fcfs' (vLLM's), 'longest_prompt_first' (waiting requests
        ranked by prompt length alone) or 'longest_first' (ranked by prompt length + max_tokens: a ranking by true
        length only for a caller that sets max_tokens per request to the length it expects)."""
        if policy not in SCHEDULING_POLICIES:
            raise ValueError("scheduling_policy must be one of " + ", ".join(repr(p) for p in SCHEDULING_POLICIES))
        _capi.check(self._lib, self._lib.t3_engine_set_schedule_policy(self._h, SCHEDULING_POLICIES.index(policy)))
        self.scheduling_policy = policy

    def set_max_num_batched_tokens(self, positions: Optional[int]) -> None:
        """vLLM's chunked-prefill budget (the reference's engine arguments leave it at vLLM's default; its README shows
        8192): a prefill job carries at most this many prompt positions per stream and the prompt that does not fit is cut
        and continued in the next job.  None or 0 = whole prompts.  Tokens do not depend on it.  Only between jobs."""
        _capi.check(self._lib, self._lib.t3_engine_set_prefill_chunk(self._h, int(positions or 0)))
        self.max_num_batched_tokens = int(positions) if positions else None

    def set_timing(self, on: bool) -> None:
        _capi.check(self._lib, self._lib.t3_engine_set_timing(self._h, int(on)))

    def set_kernel_timing(self, on: bool) -> None:
        """Time every kernel launch with HIP events bound to its dispatch (eager launches while on)."""
        _capi.check(self._lib, self._lib.t3_engine_set_kernel_timing(self._h, int(on)))

    def kernel_times(self, phase: str = "decode") -> Dict[str, dict]:
        """{class: {"ms": total, "launches": n, "bytes": algorithmic bytes}} since timing was switched on."""
        n = self._lib.t3_kernel_class_count()
        ms, cnt, byt = (C.c_double * n)(), (C.c_int64 * n)(), (C.c_int64 * n)()
        _capi.check(self._lib, self._lib.t3_engine_get_kernel_times(self._h, 0 if phase == "decode" else 1, ms, cnt, byt, n))
        return {self._lib.t3_kernel_class_name(i).decode(): {"ms": ms[i], "launches": cnt[i], "bytes": byt[i]}
                for i in range(n) if cnt[i]}

    def set_keep_logits(self, on: bool) -> None:
        _capi.check(self._lib, self._lib.t3_engine_set_keep_logits(self._h, int(on)))

    def last_logits(self, request_id: int) -> torch.Tensor:
        out = torch.empty(self.model_config.speech_vocab + SPEECH_TOKEN_OFFSET, dtype=torch.float32)
        _capi.check(self._lib, self._lib.t3_engine_last_logits(
            self._h, int(request_id), C.cast(_capi.ptr(out), C.POINTER(C.c_float))))
        return out

    # ---- the LLM.generate stand-in ---------------------------------------------------
    def _tokenize(self, item: dict) -> List[int]:
        if "prompt_token_ids" in item:
            return [int(t) for t in item["prompt_token_ids"]]
        if self.tokenizer is None:
            raise ValueError("engine has no tokenizer: pass prompt_token_ids")
        kw = item.get("tokenization_kwargs") or {}
        return list(self.tokenizer(item["prompt"], **kw))

    def generate(self, request_items: Union[dict, Sequence[dict]],
                 sampling_params: Union[SamplingParams, Sequence[SamplingParams], None] = None,
                 on_step: Optional[Callable[["T3Engine"], None]] = None) -> List[RequestOutput]:
        if isinstance(request_items, dict):
            request_items = [request_items]
        n = len(request_items)
        if sampling_params is None:
            sampling_params = SamplingParams()
        sps = list(sampling_params) if isinstance(sampling_params, (list, tuple)) else [sampling_params] * n
        if len(sps) != n:
            raise ValueError("one SamplingParams per request item is required")
        with self._generate_lock:
            return self._generate_locked(request_items, sps, on_step)

    def _generate_locked(self, request_items, sps, on_step) -> List[RequestOutput]:
        outs: List[RequestOutput] = []
        by_id: Dict[int, RequestOutput] = {}
        added: List[int] = []
        try:
            for i, (item, sp) in enumerate(zip(request_items, sps)):
                ids = self._tokenize(item)
                conds = (item.get("multi_modal_data") or {}).get("conditionals")
                assert conds is not None and len(conds) > 0, "Conditionals are required for prefill"   # t3.py:248
                assert len(conds) == 1, "Only one conditional embedding is supported for prefill"       # t3.py:249
                rid = self.add_request(ids, conds[0], sp)
                added.append(rid)
                ro = RequestOutput(request_id=str(rid), prompt=item.get("prompt"), prompt_token_ids=ids,
                                   outputs=[CompletionOutput(index=0, token_ids=[])], finished=False)
                outs.append(ro)
                by_id[rid] = ro
        except Exception:
            for rid in added:
                self.abort(rid)
            raise
        try:
            while self.has_unfinished():
                for rid, tok, fin in self.step():
                    ro = by_id.get(rid)
                    if ro is None:
                        continue
                    if fin != "error":
                        ro.outputs[0].token_ids.append(tok)
                    if fin is not None:
                        ro.outputs[0].finish_reason = fin
                        ro.finished = True
                if on_step is not None:
                    on_step(self)
        except BaseException:
            # an interrupted call must not leave its requests behind for the next one: abort them and let the
            # steps in flight drain (their tokens are dropped)
            for rid, ro in by_id.items():
                if not ro.finished:
                    try:
                        self.abort(rid)
                    except Exception:
                        pass
            try:
                while self.has_unfinished():
                    self.step()
            except Exception:
                pass
            raise
        bad = [ro.request_id for ro in outs if ro.outputs[0].finish_reason == "error"]
        if bad:
            # every other request of the call ran to its end; the error carries all outputs (the failed ones with the
            # tokens they had and finish_reason "error") for a caller that wants the healthy ones
            err = _capi.T3Error(f"request(s) {bad}: logits without a finite maximum (non-finite weights or conditioning)")
            err.outputs = outs
            raise err
        return outs

    # ---- life cycle -------------------------------------------------------------------
    def shutdown(self) -> None:
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.t3_engine_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.shutdown()
        except Exception:
            pass
