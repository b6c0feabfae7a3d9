#!/usr/bin/env python3
"""Measures how far the engine's logits are from the oracle's, by depth, and writes the distribution as JSON:

    python3 tools/logit_error_by_depth.py > gpurun_out/r02_logit_error_by_depth.json      (on the GPU box)

For each checkpoint (flat = synthetic_state_dict, decisive = decisive_state_dict), layer count and CFG scale: three
requests x 24 greedy tokens through the HIP engine with logit taps, the CPU oracle teacher-forced on the engine's ids,
and over all finite penalised logits of all steps: quantiles of |engine - oracle|, its worst value, the worst value
in units of the stated tolerance (tests/parity_bar.py: tol_at(logit, logit_tol(cfg_scale, L), head-row scale)), and the share of
steps on which both picked the same id.  The tolerance's depth factor max(1, sqrt(L/8)) is checked against this file."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from chatterbox_vllm_amd.config import T3ModelConfig, SPEECH_TOKEN_OFFSET, STOP_SPEECH_TOKEN   # noqa: E402
from chatterbox_vllm_amd import weights as W                                                    # noqa: E402
from chatterbox_vllm_amd.engine import T3Engine, SamplingParams                                 # noqa: E402
from oracle.t3_oracle import T3Oracle, SamplingSpec                                              # noqa: E402
import parity_bar as PB                                                                          # noqa: E402

N_TOK = 24


def text(n, seed):
    g = torch.Generator().manual_seed(seed)
    return [255] + torch.randint(3, 704, (n,), generator=g).tolist() + [0]


def measure(kind, L, s):
    cfg = T3ModelConfig.english(n_layers=L)
    sd = (W.decisive_state_dict if kind == "decisive" else W.synthetic_state_dict)(cfg)
    texts, voices = [text(6, 31), text(25, 32), text(70, 33)], [0, 1, 2]
    eng = T3Engine(cfg, sd, max_model_len=256, max_num_seqs=4, cfg_scale=s)
    eng.set_keep_logits(True)
    sp = SamplingParams(temperature=0.0, repetition_penalty=2.0, max_tokens=N_TOK)
    rids = [eng.add_request(t, W.synthetic_cond_emb(v), sp) for t, v in zip(texts, voices)]
    ids = {r: [] for r in rids}
    logits = {r: [] for r in rids}
    while eng.has_unfinished():
        for rid, tok, fin in eng.step():
            ids[rid].append(tok)
            if fin is None:
                logits[rid].append(eng.last_logits(rid))
    eng.shutdown()
    oracle = T3Oracle(sd, n_layers=L, cfg_scale=s)
    tol = PB.logit_tol(s, L)
    row_scale = PB.head_row_scale(sd)
    errs, rel, same, steps, top_err = [], [], 0, 0, []
    for rid, t, v in zip(rids, texts, voices):
        r = oracle.generate_one(t, W.synthetic_cond_emb(v), SamplingSpec(max_tokens=N_TOK, stop_token_ids=()), max_model_len=256,
                                forced=ids[rid], return_logits=True)
        for k, row in enumerate(logits[rid]):
            ref = r["logits"][k]
            fin = torch.isfinite(ref)
            d = (row[fin] - ref[fin]).abs()
            errs.append(d)
            rel.append((d / PB.tol_at(ref[fin], tol, row_scale[fin])).max().item())
            top_err.append(abs(row[ids[rid][k]].item() - ref[ids[rid][k]].item()))
            same += int(ids[rid][k] == r["picks"][k])
            steps += 1
    e = torch.cat(errs)
    q = torch.quantile(e[torch.randperm(e.numel())[:2_000_000]], torch.tensor([0.5, 0.9, 0.99, 0.999]))
    return {"checkpoint": kind, "layers": L, "cfg_scale": s, "steps": steps, "logits_compared": int(e.numel()),
            "abs_err_p50": round(q[0].item(), 5), "abs_err_p90": round(q[1].item(), 5), "abs_err_p99": round(q[2].item(), 5),
            "abs_err_p999": round(q[3].item(), 5), "abs_err_max": round(e.max().item(), 5),
            "tolerance_below_8": round(tol, 5), "worst_in_tolerances": round(max(rel), 3),
            "worst_err_of_the_chosen_logit": round(max(top_err), 5),
            "same_pick_share": round(same / steps, 3)}


def main():
    out = []
    for kind in ("flat", "decisive"):
        for L in (2, 8, 16, 30):
            for s in (0.5, 3.0):
                rec = measure(kind, L, s)
                print(json.dumps(rec), file=sys.stderr, flush=True)
                out.append(rec)
    json.dump({"what": "engine vs oracle, |difference| of the CFG-combined penalised fp32 logits; see tools/logit_error_by_depth.py",
               "depth_factor": "max(1, sqrt(L/8))", "cases": out}, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
