#!/usr/bin/env python3
"""Writes tests/golden/oracle_golden.json: what the CPU oracle (oracle/t3_oracle.py) computes for small, seeded
cases, so that (a) a change of the oracle's arithmetic shows up as a diff of committed data and (b) the GPU engine
can be checked against data in the tree.

    python tests/golden/make_oracle_golden.py            (a few minutes on 8 cores)

Flat cases (weights: chatterbox_vllm_amd.weights.synthetic_state_dict, seed 1234; cond: synthetic_cond_emb(voice)):
  micro  2 layers, two requests, 16 greedy tokens, CFG scale 0.5
  full   30 layers, two requests (text ids of length 7 + [START]/[STOP]), 16 greedy tokens, CFG scale 0.5
Their logits are nearly flat (top-2 gaps of a few bf16 ulp), so they serve as LOGIT checks: ids are compared only on
the steps that happen to be decisive.

Decisive cases (weights: decisive_state_dict, seed 1234):
  decisive_micro  2 layers,  >= 4 requests, up to 80 greedy tokens, CFG scale 0.5
  decisive_full   30 layers, >= 4 requests, up to 80 greedy tokens, CFG scale 0.5
Requests are searched among seeded candidates (text length, voice) and kept only if EVERY step of EVERY stream mode
has a top-2 margin above DECISIVE_FACTOR tolerances (tests/parity_bar.py), its first token differs from those of the
requests already kept (so the walks differ) and the request yields at least 64 tokens;
one more request that ends early on the stop id is kept when the search meets one.  This script refuses to write a
decisive case with fewer than four such requests.  For these cases the tests require engine ids == golden ids for
the whole sequence.

Every case is run under the three semantics of the uncond stream (own KV / no attention / CFG off).  Per request
and mode: the prompt ids, the greedy ids (+1000 space), the finish reason, per step the top-2 margin of the penalised
logits, the top logit, the first 8 speech logits and (decisive cases) the larger head-row scale of the top-2 columns.
"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from chatterbox_vllm_amd.config import T3ModelConfig            # noqa: E402
from chatterbox_vllm_amd import weights as W                    # noqa: E402
from oracle.t3_oracle import T3Oracle, SamplingSpec, build_prompt_ids   # noqa: E402
import parity_bar as PB                                         # noqa: E402

CFG_SCALE = 0.5
N_TOK = 16
DEC_TOK = 80
DEC_MIN = 64
MODES = (("paired_own_kv", dict(uncond_attends=True), True),
         ("paired_uncond_no_attention", dict(uncond_attends=False), True),
         ("single_stream", dict(), False))


def text(n, seed):
    g = torch.Generator().manual_seed(seed)
    return [255] + torch.randint(3, 704, (n,), generator=g).tolist() + [0]


def record(r, row_scale=None):
    L = r["logits"]
    top2, idx = torch.topk(L, 2, dim=-1)
    rec = {} if row_scale is None else {"top2_row_scale": [round(float(row_scale[i].max()), 4) for i in idx]}
    return {**rec, "token_ids": r["token_ids"], "finish_reason": r["finish_reason"],
            "margins": [round(float(a - b), 6) for a, b in top2],
            "top_logit": [round(float(a), 6) for a, _ in top2],
            "logits_1000_1008": [[round(float(x), 6) for x in row[1000:1008]] for row in L]}


def case(n_layers, texts, voices):
    cfg = T3ModelConfig.english(n_layers=n_layers)
    sd = W.synthetic_state_dict(cfg)
    out = {"n_layers": n_layers, "cfg_scale": CFG_SCALE, "max_tokens": N_TOK, "max_model_len": 128, "requests": []}
    for t, v in zip(texts, voices):
        rec = {"text_ids": t, "voice": v, "prompt_ids": build_prompt_ids(t), "modes": {}}
        for mode, kw, paired in MODES:
            o = T3Oracle(sd, n_layers=n_layers, cfg_scale=CFG_SCALE if paired else 0.0, **kw)
            r = o.generate_one(t, W.synthetic_cond_emb(v), SamplingSpec(max_tokens=N_TOK), max_model_len=128,
                               paired=paired, return_logits=True)
            rec["modes"][mode] = record(r)
        out["requests"].append(rec)
    return out


def all_decisive(r, tol, row_scale):
    return all(PB.step_is_decisive(row, tol, row_scale) for row in r["logits"])


def decisive_case(n_layers, want=4, max_candidates=400):
    """Search seeded candidates for requests whose every step is decisive in every mode."""
    cfg = T3ModelConfig.english(n_layers=n_layers)
    sd = W.decisive_state_dict(cfg)
    row_scale = PB.head_row_scale(sd)
    oracles = {mode: T3Oracle(sd, n_layers=n_layers, cfg_scale=CFG_SCALE if paired else 0.0, **kw)
               for mode, kw, paired in MODES}
    tols = {mode: PB.logit_tol(CFG_SCALE if paired else 0.0, n_layers) for mode, _, paired in MODES}
    out = {"n_layers": n_layers, "cfg_scale": CFG_SCALE, "max_tokens": DEC_TOK, "max_model_len": 256,
           "checkpoint": "decisive_state_dict(seed=1234)", "decisive_factor": PB.DECISIVE_FACTOR, "requests": []}
    long_ones, short_stop, tried = 0, 0, 0
    first_tokens = set()
    t0 = time.time()
    for cand in range(max_candidates):
        if long_ones >= want and (short_stop >= 1 or cand > 120):
            break
        t = text((3, 9, 21, 38, 57)[cand % 5], 9000 + cand)
        v = cand % 3
        cond = W.synthetic_cond_emb(v)
        tried += 1
        # stage 1: the first step (the only one that depends on the whole prompt) in every mode; requests whose walk
        # starts where an accepted one started would only repeat it
        ok = True
        for mode, _, paired in MODES:
            r = oracles[mode].generate_one(t, cond, SamplingSpec(max_tokens=1), max_model_len=256, paired=paired,
                                           return_logits=True)
            if not all_decisive(r, tols[mode], row_scale) or r["token_ids"][0] in first_tokens:
                ok = False
                break
        if not ok:
            continue
        # stage 2: the whole sequence
        rec = {"text_ids": t, "voice": v, "prompt_ids": build_prompt_ids(t), "modes": {}}
        for mode, _, paired in MODES:
            r = oracles[mode].generate_one(t, cond, SamplingSpec(max_tokens=DEC_TOK), max_model_len=256, paired=paired,
                                           return_logits=True)
            if not all_decisive(r, tols[mode], row_scale):
                ok = False
                break
            rec["modes"][mode] = record(r, row_scale)
        if not ok:
            continue
        lens = [len(m["token_ids"]) for m in rec["modes"].values()]
        if min(lens) >= DEC_MIN and long_ones < want:
            long_ones += 1
        elif max(lens) < DEC_MIN and min(lens) >= 4 and short_stop < 1:
            short_stop += 1                                  # ends on the stop id early: covers the stop rule
        else:
            continue
        out["requests"].append(rec)
        first_tokens.update(m["token_ids"][0] for m in rec["modes"].values())
        print(f"  L={n_layers}: kept candidate {cand} (text {len(t)}, voice {v}), lengths {lens}, "
              f"{time.time() - t0:.0f} s", flush=True)
    if long_ones < want:
        raise SystemExit(f"only {long_ones} decisive requests of >= {DEC_MIN} tokens among {tried} candidates at "
                         f"{n_layers} layers: not writing a golden")
    out["candidates_tried"] = tried
    return out


def main():
    torch.set_num_threads(8)
    data = {"micro": case(2, [text(5, 11), text(40, 12)], [0, 1]),
            "full": case(30, [text(7, 1), text(7, 2)], [0, 1]),
            "decisive_micro": decisive_case(2),
            "decisive_full": decisive_case(30)}
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_golden.json")
    with open(path, "w") as f:
        json.dump(data, f, indent=0)
        f.write("\n")
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
