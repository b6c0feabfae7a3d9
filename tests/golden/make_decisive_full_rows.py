"""Writes tests/golden/decisive_full_rows.npz: for one request of the 30-layer decisive golden per stream mode, all 8194
speech logits of every eighth decode step, computed by the CPU oracle (oracle/t3_oracle.py) on the decisive checkpoint.
The ids the oracle produces on the way must equal those already committed in oracle_golden.json.
Run from the repository root: python tests/golden/make_decisive_full_rows.py (about a minute on 8 cores)."""
import json, os, sys, numpy as np, torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from chatterbox_vllm_amd.config import T3ModelConfig
from chatterbox_vllm_amd import weights as W
from oracle.t3_oracle import T3Oracle, SamplingSpec
torch.set_num_threads(8)
gold = json.load(open(os.path.join(ROOT, "tests", "golden", "oracle_golden.json")))["decisive_full"]
L = gold["n_layers"]
sd = W.decisive_state_dict(T3ModelConfig.english(n_layers=L))
out = {}
steps = list(range(0, 80, 8))
for k, mode, kw, paired in ((1, "paired_own_kv", dict(uncond_attends=True), True), (2, "paired_uncond_no_attention", dict(uncond_attends=False), True),
                            (3, "single_stream", dict(), False)):
    req = gold["requests"][k]
    o = T3Oracle(sd, n_layers=L, cfg_scale=gold["cfg_scale"] if paired else 0.0, **kw)
    r = o.generate_one(req["text_ids"], W.synthetic_cond_emb(req["voice"]), SamplingSpec(max_tokens=gold["max_tokens"]),
                       max_model_len=gold["max_model_len"], paired=paired, return_logits=True)
    assert r["token_ids"] == req["modes"][mode]["token_ids"]
    out[f"{mode}_request"] = np.int32(k)
    out[f"{mode}_steps"] = np.array(steps, dtype=np.int32)
    out[f"{mode}_rows"] = torch.stack([r["logits"][s][1000:] for s in steps]).numpy()
    print(mode, k, len(r["token_ids"]), flush=True)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "decisive_full_rows.npz"), **out)
print(os.path.getsize(os.path.join(ROOT, "tests", "golden", "decisive_full_rows.npz")))
