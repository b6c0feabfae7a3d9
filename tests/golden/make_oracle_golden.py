#!/usr/bin/env python3
"""Writes tests/golden/oracle_golden.json: what the CPU oracle (oracle/t3_oracle.py) computes for two small,
seeded cases, so that (a) a change of the oracle's arithmetic shows up as a diff of committed data and (b) the GPU
engine can be checked against data in the tree.

    python tests/golden/make_oracle_golden.py

Cases (weights: chatterbox_vllm_amd.weights.synthetic_state_dict, seed 1234; cond: synthetic_cond_emb(voice)):
  micro  2 layers, two requests, 16 greedy tokens, CFG scale 0.5
  full   30 layers, two requests (text ids of length 7 + [START]/[STOP]), 16 greedy tokens, CFG scale 0.5
each under both semantics of the uncond stream (own KV / no attention), plus the single-stream (CFG off) ids.
Per request and mode: the prompt ids, the greedy ids (+1000 space), per step the top-2 margin of the penalised
logits, and the first 8 speech logits of every step.
"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from chatterbox_vllm_amd.config import T3ModelConfig            # noqa: E402
from chatterbox_vllm_amd import weights as W                    # noqa: E402
from oracle.t3_oracle import T3Oracle, SamplingSpec, build_prompt_ids   # noqa: E402

CFG_SCALE = 0.5
N_TOK = 16


def text(n, seed):
    g = torch.Generator().manual_seed(seed)
    return [255] + torch.randint(3, 704, (n,), generator=g).tolist() + [0]


def case(n_layers, texts, voices):
    cfg = T3ModelConfig.english(n_layers=n_layers)
    sd = W.synthetic_state_dict(cfg)
    out = {"n_layers": n_layers, "cfg_scale": CFG_SCALE, "max_tokens": N_TOK, "max_model_len": 128, "requests": []}
    for t, v in zip(texts, voices):
        rec = {"text_ids": t, "voice": v, "prompt_ids": build_prompt_ids(t), "modes": {}}
        for mode, kw, paired in (("paired_own_kv", dict(uncond_attends=True), True),
                                 ("paired_uncond_no_attention", dict(uncond_attends=False), True),
                                 ("single_stream", dict(), False)):
            o = T3Oracle(sd, n_layers=n_layers, cfg_scale=CFG_SCALE if paired else 0.0, **kw)
            r = o.generate_one(t, W.synthetic_cond_emb(v), SamplingSpec(max_tokens=N_TOK), max_model_len=128,
                               paired=paired, return_logits=True)
            L = r["logits"]
            top2 = torch.topk(L, 2, dim=-1).values
            rec["modes"][mode] = {
                "token_ids": r["token_ids"],
                "margins": [round(float(a - b), 6) for a, b in top2],
                "logits_1000_1008": [[round(float(x), 6) for x in row[1000:1008]] for row in L],
            }
        out["requests"].append(rec)
    return out


def main():
    torch.set_num_threads(8)
    data = {"micro": case(2, [text(5, 11), text(40, 12)], [0, 1]),
            "full": case(30, [text(7, 1), text(7, 2)], [0, 1])}
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_golden.json")
    with open(path, "w") as f:
        json.dump(data, f, indent=0)
        f.write("\n")
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
