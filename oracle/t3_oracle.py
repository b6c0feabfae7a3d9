"""CPU oracle for the T3 speech-token decode path.  TEST INFRASTRUCTURE ONLY.

This file is the checker the HIP engine is compared against.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import
it; the product package never does.

PARITY STATUS: *unpinned against the real vLLM execution.*  The reference delegates
the Llama layers, paged attention and the sampler to the ``vllm==0.10.0`` wheel
(pyproject.toml:29), which is neither under /root/reference nor installable here,
and the reference ships no tests or golden vectors.  What *is* pinned:

* prompt layout, embedding assembly and the CFG/offset logits of the reference's own
  ``T3VllmModel`` -- checked against outputs captured from the reference functions
  themselves (tests/golden/make_reference_fixtures.py -> ref_plugin_fixture.npz);
* the Llama block -- checked against ``transformers`` ``LlamaModel`` built from the
  reference's config dict (tests/test_oracle.py).

Restated from (paths relative to /root/reference):
  prompt layout ............ src/chatterbox_vllm/models/t3/t3.py:229-240
  prefill embeddings ....... t3.py:548-569   (cond | text+pos | SOS+pos0, uncond text = 0)
  decode embeddings ........ t3.py:503-514   (speech_emb[id-1000], no position table)
  stream pairing ........... t3.py:762-778   (cond and uncond rows share positions)
  CFG + 1000 -inf columns .. t3.py:725-739
  backbone hyper-params .... t3-model/config.json:8-32
  sampling parameters ...... src/chatterbox_vllm/tts.py:408-418
  stop / length rules ...... tts.py:402, 448-456 (stop id is kept in the output)

Arithmetic contract in ``bf16=True`` mode (storage bf16, accumulate fp32, one rounding
per op output): residual add -> bf16; RMSNorm = bf16(bf16(x*rsqrt(mean(x^2)+eps))*w);
linear outputs -> bf16; RoPE (half-split pairs) from a bf16 cos/sin table, one rounding;
attention scale 1/8, fp32 softmax and PV, output -> bf16; SiLU in fp32 -> bf16, then
bf16 multiply by up; head logits -> bf16; CFG ``c + s*(c-u)`` with three bf16
roundings; sampler in fp32.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import torch

# t3.py:38-49, modules/t3_config.py:8-11
COND_START, COND_END, PREFILL_END = 695, 696, 697
N_COND = 34
OFFSET = 1000
SOS, EOS = 6561, 6562
SPEECH_VOCAB = 8194


def build_prompt_ids(text_ids: Sequence[int]) -> List[int]:
    """t3.py:229-240 -- [695, ids[0] x 32, 696, *ids, 697]."""
    text_ids = list(text_ids)
    return [COND_START, *([text_ids[0]] * (N_COND - 2)), COND_END, *text_ids, PREFILL_END]


def llama3_inv_freq(head_dim=64, theta=500000.0, factor=8.0, low=1.0, high=4.0, orig=8192):
    """t3-model/config.json:21-28 (rope_type llama3)."""
    # fp32 tensor arithmetic throughout, as the serving stack computes it: a float64
    # evaluation differs in the last fp32 bit for some bands and flips bf16 table entries.
    inv = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.float32) / head_dim))
    wl = 2 * math.pi / inv
    smooth = (orig / wl - low) / (high - low)
    return torch.where(wl < orig / high, inv,
                       torch.where(wl > orig / low, inv / factor,
                                   (1 - smooth) * inv / factor + smooth * inv)).float()


def make_rope_table(max_pos: int, **kw) -> torch.Tensor:
    """bf16 [max_pos, 64]: cos | sin, angles in fp32 then cast to the model dtype."""
    inv = llama3_inv_freq(**kw)
    ang = torch.arange(max_pos, dtype=torch.float32)[:, None] * inv[None, :]
    return torch.cat([ang.cos(), ang.sin()], dim=1).to(torch.bfloat16)


@dataclass
class SamplingSpec:
    """The subset of vLLM SamplingParams the reference sets (tts.py:410-417)."""
    temperature: float = 0.0
    top_p: float = 1.0
    min_p: float = 0.0
    repetition_penalty: float = 2.0
    max_tokens: int = 1000
    stop_token_ids: Tuple[int, ...] = (EOS + OFFSET,)


@dataclass
class _Stream:
    """One KV stream (cond or uncond half of a request)."""
    k: List[torch.Tensor] = field(default_factory=list)   # per layer [S,H,hd]
    v: List[torch.Tensor] = field(default_factory=list)
    attends: bool = True
    # fault injection for the golden maker: (layer, first position, end position) whose K and V read as zero, the way
    # a lost or never-written KV page would; None = no fault
    lost_kv: Optional[Tuple[int, int, int]] = None

    def apply_fault(self, li: int) -> None:
        if self.lost_kv is not None and self.lost_kv[0] == li:
            _, a, b = self.lost_kv
            self.k[li][a:b] = 0
            self.v[li][a:b] = 0


class T3Oracle:
    def __init__(self, state_dict: Dict[str, torch.Tensor], n_layers: int = 30,
                 n_heads: int = 16, head_dim: int = 64, rms_eps: float = 1e-5,
                 cfg_scale: float = 0.5, bf16: bool = True, uncond_attends: bool = True,
                 rope: Optional[torch.Tensor] = None, max_pos: int = 4096, accumulate_fp64: bool = False):
        self.L, self.H, self.hd = n_layers, n_heads, head_dim
        # accumulate_fp64: matrix products summed in double before the one rounding.  Same contract, other summation
        # error: the golden maker uses it to reject cases whose ids hinge on the order of an fp32 sum.
        self.accumulate_fp64 = accumulate_fp64
        self.eps = rms_eps
        self.cfg_scale = float(cfg_scale)
        self.bf16 = bf16
        self.uncond_attends = uncond_attends
        f = lambda n: state_dict[n].to(torch.float32)
        self.text_emb = f("text_emb.weight")
        self.speech_emb = f("speech_emb.weight")
        self.text_pos = f("text_pos_emb.emb.weight")
        self.speech_pos = f("speech_pos_emb.emb.weight")
        self.head = f("speech_head.weight")
        self.norm_w = f("tfmr.norm.weight")
        self.layers = []
        for i in range(n_layers):
            p = f"tfmr.layers.{i}."
            self.layers.append(dict(
                ln1=f(p + "input_layernorm.weight"),
                wqkv=torch.cat([f(p + "self_attn.q_proj.weight"), f(p + "self_attn.k_proj.weight"),
                                f(p + "self_attn.v_proj.weight")], 0),
                wo=f(p + "self_attn.o_proj.weight"),
                ln2=f(p + "post_attention_layernorm.weight"),
                wgu=torch.cat([f(p + "mlp.gate_proj.weight"), f(p + "mlp.up_proj.weight")], 0),
                wd=f(p + "mlp.down_proj.weight"),
            ))
        self.d = self.text_emb.shape[1]
        self.ff = self.layers[0]["wd"].shape[1] if n_layers else 0
        if rope is None:
            rope = make_rope_table(max_pos, head_dim=head_dim)
        rope = rope.to(torch.float32) if bf16 else self._fp32_rope(rope.shape[0])
        self.cos, self.sin = rope[:, : head_dim // 2], rope[:, head_dim // 2:]

    def _fp32_rope(self, max_pos):
        inv = llama3_inv_freq(head_dim=self.hd)
        ang = torch.arange(max_pos, dtype=torch.float32)[:, None] * inv[None, :]
        return torch.cat([ang.cos(), ang.sin()], dim=1)

    # ---- rounding helper ------------------------------------------------------
    def r(self, x: torch.Tensor) -> torch.Tensor:
        return x.to(torch.bfloat16).to(torch.float32) if self.bf16 else x

    def mm(self, x: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
        """x @ w.T, rounded once to the storage type."""
        if self.accumulate_fp64:
            return self.r((x.double() @ w.double().T).float())
        return self.r(x @ w.T)

    # ---- embeddings -----------------------------------------------------------
    def prefill_embeds(self, text_ids: Sequence[int], cond_emb: torch.Tensor
                       ) -> Tuple[torch.Tensor, torch.Tensor]:
        """t3.py:548-569 for a whole prompt.  Returns (cond, uncond) [34+T+1, d]."""
        ids = torch.tensor(list(text_ids), dtype=torch.long)
        T = len(ids)
        cond_rows = self.r(cond_emb.to(torch.float32))                  # fp32 -> model dtype at copy
        text = self.r(self.text_emb[ids] + self.text_pos[:T])            # bf16 + bf16 -> bf16
        sos = self.r(self.speech_emb[SOS:SOS + 1] + self.speech_pos[0:1])
        cond = torch.cat([cond_rows, text, sos], 0)
        uncond = torch.cat([cond_rows, torch.zeros_like(text), sos], 0)
        return cond, uncond

    def decode_embed(self, token_plus_offset: int) -> torch.Tensor:
        """t3.py:503-508 -- speech_emb[id-1000] for both halves, no position table."""
        return self.speech_emb[token_plus_offset - OFFSET][None, :]

    # ---- backbone -------------------------------------------------------------
    def _rms(self, x, w):
        var = (x * x).mean(-1, keepdim=True)
        return self.r(self.r(x * torch.rsqrt(var + self.eps)) * w)

    def _rope(self, t, pos):
        """t [T,H,hd]; pairs (j, j+hd/2)."""
        h = self.hd // 2
        c, s = self.cos[pos][:, None, :], self.sin[pos][:, None, :]
        x, y = t[..., :h], t[..., h:]
        return self.r(torch.cat([x * c - y * s, y * c + x * s], -1))

    def forward(self, x: torch.Tensor, pos: torch.Tensor, st: _Stream,
                taps: Optional[dict] = None) -> torch.Tensor:
        """x [T,d] (already rounded embeddings), pos [T] absolute positions.  Appends this
        chunk's K/V to ``st`` and returns the final-normed hidden states [T,d]."""
        T = x.shape[0]
        res = None
        new_stream = len(st.k) == 0
        for li, lw in enumerate(self.layers):
            if res is None:
                res = x
            else:
                res = self.r(x + res)
            xn = self._rms(res, lw["ln1"])
            qkv = self.mm(xn, lw["wqkv"])
            q, k, v = qkv.split(self.d, dim=-1)
            q = self._rope(q.view(T, self.H, self.hd), pos)
            k = self._rope(k.view(T, self.H, self.hd), pos)
            v = v.reshape(T, self.H, self.hd)
            if new_stream:
                st.k.append(k), st.v.append(v)
            else:
                st.k[li] = torch.cat([st.k[li], k], 0)
                st.v[li] = torch.cat([st.v[li], v], 0)
            st.apply_fault(li)
            if st.attends:
                K, V = st.k[li], st.v[li]
                S = K.shape[0]
                kpos = torch.arange(S)  # stream positions are contiguous from 0
                sc = torch.einsum("thd,shd->hts", q, K) * 0.125
                mask = kpos[None, :] > pos[:, None]
                sc = sc.masked_fill(mask[None], float("-inf"))
                p = torch.softmax(sc, dim=-1)
                a = self.r(torch.einsum("hts,shd->thd", p, V)).reshape(T, self.d)
            else:
                a = torch.zeros(T, self.d)
            x = self.mm(a, lw["wo"])
            res = self.r(x + res)
            xn = self._rms(res, lw["ln2"])
            gu = self.mm(xn, lw["wgu"])
            g, u = gu.split(self.ff, dim=-1)
            h = self.r(self.r(g / (1.0 + torch.exp(-g))) * u)
            x = self.mm(h, lw["wd"])
            if taps is not None and li in taps.get("layers", ()):
                taps[f"layer{li}"] = self.r(x + res).clone()
        if res is None:
            res = x
        else:
            res = self.r(x + res)
        return self._rms(res, self.norm_w)

    def forward_decode_rows(self, x: torch.Tensor, pos: Sequence[int], streams: Sequence[_Stream]) -> torch.Tensor:
        """One decode token for R different streams at once (row r belongs to streams[r] at
        position pos[r]).  Same arithmetic as ``forward`` row by row; the projections run as one
        [R, d] matmul, which is how a batched CPU implementation would serve a batch."""
        R = x.shape[0]
        pos_t = torch.tensor(list(pos), dtype=torch.long)
        res = None
        for li, lw in enumerate(self.layers):
            res = x if res is None else self.r(x + res)
            xn = self._rms(res, lw["ln1"])
            qkv = self.mm(xn, lw["wqkv"])
            q, k, v = qkv.split(self.d, dim=-1)
            q = self._rope(q.view(R, self.H, self.hd), pos_t)
            k = self._rope(k.view(R, self.H, self.hd), pos_t)
            v = v.reshape(R, self.H, self.hd)
            rows = []
            for r_, st in enumerate(streams):
                st.k[li] = torch.cat([st.k[li], k[r_:r_ + 1]], 0)
                st.v[li] = torch.cat([st.v[li], v[r_:r_ + 1]], 0)
                st.apply_fault(li)
                if st.attends:
                    sc = torch.einsum("hd,shd->hs", q[r_], st.k[li]) * 0.125
                    rows.append(torch.einsum("hs,shd->hd", torch.softmax(sc, -1), st.v[li]).reshape(self.d))
                else:
                    rows.append(torch.zeros(self.d))
            a = self.r(torch.stack(rows))
            x = self.mm(a, lw["wo"])
            res = self.r(x + res)
            xn = self._rms(res, lw["ln2"])
            gu = self.mm(xn, lw["wgu"])
            g, u = gu.split(self.ff, dim=-1)
            x = self.mm(self.r(self.r(g / (1.0 + torch.exp(-g))) * u), lw["wd"])
        res = x if res is None else self.r(x + res)
        return self._rms(res, self.norm_w)

    def generate_batch(self, requests: Sequence[Tuple[Sequence[int], torch.Tensor]], out_lens: Sequence[int],
                       repetition_penalty: float = 2.0, paired: bool = True) -> List[List[int]]:
        """Greedy ids for a batch served together: per-request prefill, then batched decode steps
        (rows = cond streams then uncond streams).  Used as the timed CPU baseline."""
        n = len(requests)
        sc = [_Stream() for _ in range(n)]
        su = [_Stream(attends=self.uncond_attends) for _ in range(n)]
        hc, hu, P, seen, out = [], [], [], [], [[] for _ in range(n)]
        for i, (text, cond) in enumerate(requests):
            ce, ue = self.prefill_embeds(text, cond)
            p = torch.arange(ce.shape[0])
            hc.append(self.forward(ce, p, sc[i])[-1])
            hu.append(self.forward(ue, p, su[i])[-1] if paired else None)
            P.append(ce.shape[0])
            seen.append(list(build_prompt_ids(text)))
        active = list(range(n))
        while active:
            lc = self.head_logits(torch.stack([hc[i] for i in active]))
            lu = self.head_logits(torch.stack([hu[i] for i in active])) if paired else None
            logits = self.cfg_logits(lc, lu)
            for j, i in enumerate(active):
                tok = self.greedy(self.apply_repetition_penalty(logits[j], seen[i], repetition_penalty))
                out[i].append(tok)
                seen[i].append(tok)
            active = [i for i in active if len(out[i]) < out_lens[i]]
            if not active:
                break
            e = self.r(torch.cat([self.decode_embed(out[i][-1]) for i in active]))
            pos = [P[i] + len(out[i]) - 1 for i in active]
            if paired:
                h = self.forward_decode_rows(torch.cat([e, e]), pos + pos, [sc[i] for i in active] + [su[i] for i in active])
                for j, i in enumerate(active):
                    hc[i], hu[i] = h[j], h[len(active) + j]
            else:
                h = self.forward_decode_rows(e, pos, [sc[i] for i in active])
                for j, i in enumerate(active):
                    hc[i] = h[j]
        return out

    # ---- logits ---------------------------------------------------------------
    def head_logits(self, h: torch.Tensor) -> torch.Tensor:
        return self.mm(h, self.head)

    def cfg_logits(self, c: torch.Tensor, u: Optional[torch.Tensor]) -> torch.Tensor:
        """t3.py:732-738.  Returns [.., 9194]: 1000 -inf columns then the speech logits."""
        if u is None:
            l = c
        else:
            l = self.r(c + self.r(self.cfg_scale * self.r(c - u)))
        pad = torch.full((*l.shape[:-1], OFFSET), float("-inf"))
        return torch.cat([pad, l], -1)

    # ---- sampler (vLLM V1 order: penalties -> greedy | T -> min_p -> top_p -> sample)
    @staticmethod
    def apply_repetition_penalty(logits: torch.Tensor, seen: Sequence[int], p: float) -> torch.Tensor:
        l = logits.clone().float()
        if p != 1.0 and len(seen):
            idx = torch.tensor(sorted(set(int(s) for s in seen if 0 <= int(s) < l.shape[-1])), dtype=torch.long)
            vals = l[idx]
            l[idx] = torch.where(vals > 0, vals / p, vals * p)
        return l

    @staticmethod
    def greedy(l: torch.Tensor) -> int:
        return int((l == l.max()).nonzero()[0, 0])   # lowest index wins ties

    @staticmethod
    def sampling_distribution(l: torch.Tensor, spec: SamplingSpec) -> torch.Tensor:
        """Probability vector the stochastic path draws from (after penalties)."""
        l = l.float() / spec.temperature
        if spec.min_p > 0:
            pr = torch.softmax(l, -1)
            l = l.masked_fill(pr < spec.min_p * pr.max(), float("-inf"))
        if spec.top_p < 1.0:
            srt, idx = torch.sort(l, descending=False)
            cum = torch.softmax(srt, -1).cumsum(-1)
            drop = cum <= 1 - spec.top_p
            drop[-1] = False
            l = l.masked_fill(torch.zeros_like(drop).scatter(0, idx, drop), float("-inf"))
        return torch.softmax(l, -1)

    # ---- end-to-end -----------------------------------------------------------
    def generate_one(self, text_ids: Sequence[int], cond_emb: torch.Tensor, spec: SamplingSpec,
                     max_model_len: int = 1000, paired: bool = True,
                     forced: Optional[Sequence[int]] = None, return_logits: bool = False,
                     generator: Optional[torch.Generator] = None,
                     lost_kv: Optional[Tuple[int, int, int]] = None):
        """Greedy (or sampled) speech-token ids in the engine's +1000 id space, stop id
        included when emitted.  ``forced`` teacher-forces the given ids (still returns the
        oracle's own choice per step in ``picks``).  ``lost_kv`` = (layer, first, end): the cond
        stream's K/V at those positions of that layer read as zero (fault injection, see _Stream)."""
        prompt = build_prompt_ids(text_ids)
        P = len(prompt)
        if P > max_model_len:
            raise ValueError(f"prompt of {P} tokens exceeds max_model_len={max_model_len}")
        ce, ue = self.prefill_embeds(text_ids, cond_emb)
        pos = torch.arange(P)
        sc = _Stream(lost_kv=lost_kv)
        su = _Stream(attends=self.uncond_attends)
        hc = self.forward(ce, pos, sc)[-1:]
        hu = self.forward(ue, pos, su)[-1:] if paired else None
        seen = list(prompt)
        out: List[int] = []
        picks: List[int] = []
        logit_rows = []
        finish = "length"
        while True:
            lc = self.head_logits(hc)
            lu = self.head_logits(hu) if paired else None
            logits = self.cfg_logits(lc, lu)[0]
            pen = self.apply_repetition_penalty(logits, seen, spec.repetition_penalty)
            if return_logits:
                logit_rows.append(pen.clone())
            if spec.temperature < 1e-5:
                tok = self.greedy(pen)
            else:
                pr = self.sampling_distribution(pen, spec)
                q = torch.empty_like(pr).exponential_(generator=generator)
                tok = int((pr / q).argmax())
            picks.append(tok)
            if forced is not None:
                tok = int(forced[len(out)])
            out.append(tok)
            seen.append(tok)
            if tok in spec.stop_token_ids:
                finish = "stop"
                break
            if len(out) >= spec.max_tokens or P + len(out) >= max_model_len:
                break
            if forced is not None and len(out) >= len(forced):
                break
            e = self.r(self.decode_embed(tok))
            p1 = torch.tensor([P + len(out) - 1])
            hc = self.forward(e, p1, sc)
            hu = self.forward(e, p1, su) if paired else None
        res = dict(token_ids=out, finish_reason=finish, picks=picks)
        if return_logits:
            res["logits"] = torch.stack(logit_rows)
        return res

    def generate(self, requests: Sequence[Tuple[Sequence[int], torch.Tensor]], spec: SamplingSpec,
                 max_model_len: int = 1000, paired: bool = True) -> List[dict]:
        return [self.generate_one(t, c, spec, max_model_len, paired) for t, c in requests]


def postprocess_ids(token_ids: Sequence[int], tail_crop: int = 2) -> List[int]:
    """tts.py:448-465 plus models/s3tokenizer/__init__.py:37-79: cut at the first stop id
    and drop ``tail_crop`` more, remove the offset, keep what lies between SOS and EOS,
    keep ids < 6561."""
    ids = list(token_ids)
    stop = EOS + OFFSET
    if stop in ids:
        ids = ids[: max(0, ids.index(stop) - max(0, tail_crop))]
    ids = [t - OFFSET for t in ids]
    start = ids.index(SOS) + 1 if SOS in ids else 0      # first SOS, searched over the whole list
    end = ids.index(EOS) if EOS in ids else len(ids)     # first EOS, searched over the whole list
    ids = ids[start:end] if end > start else []
    return [t for t in ids if t < SOS]
