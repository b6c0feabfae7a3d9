#!/usr/bin/env python3
"""Diagnostic: the decisive goldens through the engine, printing for every mode the worst error of the chosen (top)
logit against the golden and the smallest margin / error ratio.  python3 tools/golden_error_probe.py decisive_full"""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from chatterbox_vllm_amd.config import T3ModelConfig
from chatterbox_vllm_amd import weights as W
from chatterbox_vllm_amd.engine import T3Engine, SamplingParams
import parity_bar as PB

case = sys.argv[1] if len(sys.argv) > 1 else "decisive_full"
gold = json.load(open(os.path.join(ROOT, "tests", "golden", "oracle_golden.json")))[case]
L = gold["n_layers"]
cfg = T3ModelConfig.english(n_layers=L)
sd = W.decisive_state_dict(cfg)
norms = sd["speech_head.weight"].float().norm(dim=1)
print("head row norms: median %.3f max %.3f" % (norms.median().item(), norms.max().item()))
modes = {"paired_own_kv": dict(cfg_paired=True, uncond_attends=True), "paired_uncond_no_attention": dict(cfg_paired=True, uncond_attends=False),
         "single_stream": dict(cfg_paired=False)}
reqs = gold["requests"]
for mode, kw in modes.items():
    s = gold["cfg_scale"] if kw["cfg_paired"] else 0.0
    tol = PB.logit_tol(s, L)
    eng = T3Engine(cfg, sd, max_model_len=gold["max_model_len"], max_num_seqs=len(reqs), cfg_scale=s, **kw)
    eng.set_keep_logits(True)
    sp = SamplingParams(temperature=0.0, repetition_penalty=2.0, max_tokens=gold["max_tokens"], stop_token_ids=[7562])
    rids = [eng.add_request(r["text_ids"], W.synthetic_cond_emb(r["voice"]), sp) for r in reqs]
    ids = {r: [] for r in rids}; logits = {r: [] for r in rids}
    while eng.has_unfinished():
        for rid, tok, fin in eng.step():
            ids[rid].append(tok)
            if fin is None:
                logits[rid].append(eng.last_logits(rid))
    eng.shutdown()
    for k, (rid, req) in enumerate(zip(rids, reqs)):
        g = req["modes"][mode]
        same = ids[rid] == g["token_ids"]
        worst = []
        for t, row in enumerate(logits[rid]):
            if t >= len(g["token_ids"]) or ids[rid][t] != g["token_ids"][t]:
                break
            e = abs(row[ids[rid][t]].item() - g["top_logit"][t])
            worst.append((e / max(g["margins"][t], 1e-9), e, g["margins"][t], g["top_logit"][t], norms[ids[rid][t] - 1000].item(), t))
        worst.sort(reverse=True)
        print(mode, "req", k, "ids equal" if same else "IDS DIFFER", "tol %.3f" % tol,
              "worst err/margin:", [(round(a, 3), round(b, 3), round(c, 2), round(d, 2), round(n, 2), t) for a, b, c, d, n, t in worst[:3]],
              "max err %.3f" % max(w[1] for w in worst))
