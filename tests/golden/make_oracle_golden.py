#!/usr/bin/env python3
"""Writes tests/golden/oracle_golden.json and tests/golden/long_context_golden.npz: what the CPU oracle
(oracle/t3_oracle.py) computes for small, seeded cases, so that (a) a change of the oracle's arithmetic shows up as a
diff of committed data and (b) the GPU engine can be checked against data in the tree.

    python tests/golden/make_oracle_golden.py            (about half an hour on 8 cores)

Flat cases (weights: chatterbox_vllm_amd.weights.synthetic_state_dict, seed 1234; cond: synthetic_cond_emb(voice)):
  micro  2 layers, two requests, 16 greedy tokens, CFG scale 0.5
  full   30 layers, two requests (text ids of length 7 + [START]/[STOP]), 16 greedy tokens, CFG scale 0.5
Their logits are nearly flat (top-2 gaps of a few bf16 ulp), so they serve as LOGIT checks: ids are compared only on
the steps that happen to be decisive.

Decisive cases (weights: decisive_state_dict, seed 1234: the winner of a step is picked by layer-0 attention over the
paged history at fixed look-back distances, by the text rows and by the cond / uncond pairing):
  decisive_micro  2 layers,  >= 4 requests, up to 80 greedy tokens, CFG scale 0.5
  decisive_full   30 layers, >= 4 requests, up to 80 greedy tokens, CFG scale 0.5
Requests are searched among seeded candidates (text length, voice).  A candidate is kept only if
  1. EVERY step of EVERY stream mode has a top-2 margin above DECISIVE_FACTOR tolerances (tests/parity_bar.py);
  2. its three stream modes give pairwise DIFFERENT id sequences within the first 16 steps (so a wrong pairing mode,
     a swapped cond / uncond row or an unattended uncond stream cannot reproduce the ids);
  3. losing one KV page changes the ids: with the cond stream's K/V of positions 0..63, or of 64..127, of layer 0
     reading as zero, the own-KV walk leaves the recorded one (both faults, each recorded with the step it shows at);
  4. the ids of every mode are the same when the oracle sums its matrix products in double (so no id hinges on the
     order of an fp32 sum);
  5. it yields at least 64 tokens; one more request that ends early on the stop id is kept when the search meets one.
This script refuses to write a decisive case with fewer than four such requests.  For these cases the tests require
engine ids == golden ids for the whole sequence.

Every case is run under the three semantics of the uncond stream (own KV / no attention / CFG off).  Per request
and mode: the prompt ids, the greedy ids (+1000 space), the finish reason, per step the top-2 margin of the penalised
logits, the top logit, the first 8 speech logits and (decisive cases) the larger head-row scale of the top-2 columns.

long_context_golden.npz: 30 layers, decisive_state_dict, one request whose 478 text ids make a prompt of 513 positions,
72 greedy tokens (positions 513..584: ten KV pages per stream), CFG-paired with own KV, scale 0.5.  Holds the text ids,
the ids, per step margin / top logit / row scale, and the FULL row of 8194 penalised speech logits at 10 steps; kept
only if every step is decisive, the ids survive double-precision summation, and zeroing either of the KV pages of
layer 0 that the look-back distances reach during the walk (positions 448..511 and 512..575) changes the ids.
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


def first_difference(a, b):
    return next((i for i, (x, y) in enumerate(zip(a, b)) if x != y), None if len(a) == len(b) else min(len(a), len(b)))


def decisive_case(n_layers, want=4, max_candidates=400):
    """Search seeded candidates for requests that meet rules 1-5 of the module docstring."""
    cfg = T3ModelConfig.english(n_layers=n_layers)
    sd = W.decisive_state_dict(cfg)
    row_scale = PB.head_row_scale(sd)
    oracles = {mode: T3Oracle(sd, n_layers=n_layers, cfg_scale=CFG_SCALE if paired else 0.0, **kw)
               for mode, kw, paired in MODES}
    doubles = {mode: T3Oracle(sd, n_layers=n_layers, cfg_scale=CFG_SCALE if paired else 0.0, accumulate_fp64=True, **kw)
               for mode, kw, paired in MODES}
    tols = {mode: PB.logit_tol(CFG_SCALE if paired else 0.0, n_layers) for mode, _, paired in MODES}
    out = {"n_layers": n_layers, "cfg_scale": CFG_SCALE, "max_tokens": DEC_TOK, "max_model_len": 256,
           "checkpoint": "decisive_state_dict(seed=1234)", "decisive_factor": PB.DECISIVE_FACTOR, "requests": []}
    long_ones, short_stop, tried = 0, 0, 0
    rejected = {"not decisive": 0, "modes coincide": 0, "lost page unseen": 0, "summation order": 0, "length": 0}
    t0 = time.time()
    for cand in range(max_candidates):
        if long_ones >= want and (short_stop >= 1 or cand > 120):
            break
        t = text((3, 9, 21, 38, 57)[cand % 5], 9000 + cand)
        v = cand % 3
        cond = W.synthetic_cond_emb(v)
        tried += 1
        rec = {"text_ids": t, "voice": v, "prompt_ids": build_prompt_ids(t), "modes": {}}
        ids, ok = {}, True
        for mode, _, paired in MODES:                                       # rule 1
            r = oracles[mode].generate_one(t, cond, SamplingSpec(max_tokens=DEC_TOK), max_model_len=256, paired=paired,
                                           return_logits=True)
            if not all_decisive(r, tols[mode], row_scale):
                ok = False
                break
            rec["modes"][mode] = record(r, row_scale)
            ids[mode] = r["token_ids"]
        if not ok:
            rejected["not decisive"] += 1
            continue
        names = [m for m, _, _ in MODES]
        diffs = {f"{a}|{b}": first_difference(ids[a], ids[b]) for i, a in enumerate(names) for b in names[i + 1:]}
        if any(d is None or d >= 16 for d in diffs.values()):               # rule 2
            rejected["modes coincide"] += 1
            continue
        lost = {}
        for page in (0, 1):                                                 # rule 3
            r = oracles["paired_own_kv"].generate_one(t, cond, SamplingSpec(max_tokens=DEC_TOK), max_model_len=256,
                                                      lost_kv=(0, 64 * page, 64 * page + 64))
            lost[f"layer0_page{page}"] = first_difference(r["token_ids"], ids["paired_own_kv"])
        if any(d is None for d in lost.values()):
            rejected["lost page unseen"] += 1
            continue
        same = all(doubles[mode].generate_one(t, cond, SamplingSpec(max_tokens=DEC_TOK), max_model_len=256,
                                              paired=paired)["token_ids"] == ids[mode] for mode, _, paired in MODES)
        if not same:                                                        # rule 4
            rejected["summation order"] += 1
            continue
        lens = [len(x) for x in ids.values()]
        if min(lens) >= DEC_MIN and long_ones < want:                       # rule 5
            long_ones += 1
        elif min(lens) < DEC_MIN and min(lens) >= 4 and short_stop < 1:
            short_stop += 1                                  # ends on the stop id early: covers the stop rule
        else:
            rejected["length"] += 1
            continue
        rec["modes_first_differ_at"] = diffs
        rec["lost_kv_page_shows_at"] = lost
        out["requests"].append(rec)
        print(f"  L={n_layers}: kept candidate {cand} (text {len(t)}, voice {v}), lengths {lens}, modes differ at "
              f"{diffs}, lost pages show at {lost}, {time.time() - t0:.0f} s", flush=True)
    if long_ones < want:
        raise SystemExit(f"only {long_ones} decisive requests of >= {DEC_MIN} tokens among {tried} candidates at "
                         f"{n_layers} layers ({rejected}): not writing a golden")
    out["candidates_tried"] = tried
    out["candidates_rejected"] = rejected
    return out


LONG_STEPS = (0, 1, 2, 9, 19, 29, 39, 49, 59, 70)


def long_context_case(n_layers=30, n_text=478, n_tok=72, max_candidates=12):
    """One request across position 512 on the long look-back checkpoint, with full logit rows."""
    cfg = T3ModelConfig.english(n_layers=n_layers)
    sd = W.decisive_state_dict(cfg)
    row_scale = PB.head_row_scale(sd)
    tol = PB.logit_tol(CFG_SCALE, n_layers)
    o = T3Oracle(sd, n_layers=n_layers, cfg_scale=CFG_SCALE)
    o64 = T3Oracle(sd, n_layers=n_layers, cfg_scale=CFG_SCALE, accumulate_fp64=True)
    spec = SamplingSpec(max_tokens=n_tok)
    for cand in range(max_candidates):
        t, v = text(n_text - 2, 4242 + cand), cand % 3
        cond = W.synthetic_cond_emb(v)
        t0 = time.time()
        r = o.generate_one(t, cond, spec, max_model_len=640, return_logits=True)
        if len(r["token_ids"]) < n_tok or not all_decisive(r, tol, row_scale):
            print(f"  long: candidate {cand} not decisive / too short ({len(r['token_ids'])} ids)", flush=True)
            continue
        if o64.generate_one(t, cond, spec, max_model_len=640)["token_ids"] != r["token_ids"]:
            print(f"  long: candidate {cand} depends on the summation order", flush=True)
            continue
        lost = {}
        for page in (7, 8):
            f = o.generate_one(t, cond, spec, max_model_len=640, lost_kv=(0, 64 * page, 64 * page + 64))
            lost[page] = first_difference(f["token_ids"], r["token_ids"])
        print(f"  long: candidate {cand}: lost pages show at {lost}, {time.time() - t0:.0f} s", flush=True)
        if any(d is None for d in lost.values()):
            continue
        L = r["logits"]
        top2, idx = torch.topk(L, 2, dim=-1)
        return dict(n_layers=n_layers, cfg_scale=CFG_SCALE, max_tokens=n_tok, max_model_len=640, voice=v,
                    text_ids=torch.tensor(t, dtype=torch.int32).numpy(),
                    token_ids=torch.tensor(r["token_ids"], dtype=torch.int32).numpy(),
                    margins=(top2[:, 0] - top2[:, 1]).numpy(), top_logit=top2[:, 0].numpy(),
                    top2_row_scale=torch.stack([row_scale[i].max() for i in idx]).numpy(),
                    logit_steps=torch.tensor(LONG_STEPS, dtype=torch.int32).numpy(),
                    logit_rows=torch.stack([L[s, 1000:] for s in LONG_STEPS]).numpy(),
                    lost_pages=torch.tensor([7, 8], dtype=torch.int32).numpy(),
                    lost_page_shows_at=torch.tensor([lost[7], lost[8]], dtype=torch.int32).numpy())
    raise SystemExit("no long-context candidate met the rules: not writing a golden")


def main():
    import numpy as np
    torch.set_num_threads(8)
    here = os.path.dirname(os.path.abspath(__file__))
    if "--long-only" not in sys.argv:
        data = {"micro": case(2, [text(5, 11), text(40, 12)], [0, 1]),
                "full": case(30, [text(7, 1), text(7, 2)], [0, 1]),
                "decisive_micro": decisive_case(2),
                "decisive_full": decisive_case(30)}
        path = os.path.join(here, "oracle_golden.json")
        with open(path, "w") as f:
            json.dump(data, f, indent=0)
            f.write("\n")
        print("wrote", path, os.path.getsize(path), "bytes", flush=True)
    if "--json-only" not in sys.argv:
        path = os.path.join(here, "long_context_golden.npz")
        np.savez_compressed(path, **long_context_case())
        print("wrote", path, os.path.getsize(path), "bytes", flush=True)


if __name__ == "__main__":
    main()
