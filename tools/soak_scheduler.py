"""Scheduler soak: random arrivals, aborts, voices, slot counts, pool sizes, eager/graph, all three admission orders, whole and chunked prompts;
every surviving request must end with the ids of a solo run and the KV pool must drain.  python tools/soak_scheduler.py FIRST_SEED END_SEED"""
import os, sys, random, torch
sys.path.insert(0, os.getcwd())
from chatterbox_vllm_amd.config import T3ModelConfig, SPEECH_TOKEN_OFFSET, STOP_SPEECH_TOKEN
from chatterbox_vllm_amd import weights as W
from chatterbox_vllm_amd.engine import T3Engine, SamplingParams
STOP = STOP_SPEECH_TOKEN + SPEECH_TOKEN_OFFSET
cfg = T3ModelConfig.english(n_layers=2)
sd = W.synthetic_state_dict(cfg)
def text(n, seed):
    g = torch.Generator().manual_seed(seed)
    return [255] + torch.randint(3, 704, (n,), generator=g).tolist() + [0]
conds = [W.synthetic_cond_emb(v) for v in range(11)]
solo = T3Engine(cfg, sd, max_model_len=192, max_num_seqs=1, cfg_scale=3.0, prefix_sharing=False, enforce_eager=True)
cache = {}
def solo_ids(t, v, m):
    k = (tuple(t), v, m)
    if k not in cache:
        cache[k] = solo.generate({"prompt_token_ids": t, "multi_modal_data": {"conditionals": [conds[v]]}},
                                 SamplingParams(temperature=0.0, repetition_penalty=2.0, max_tokens=m, stop_token_ids=[STOP]))[0].outputs[0].token_ids
    return cache[k]
bad = 0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    rng = random.Random(seed)
    n = 40
    texts = [text(rng.choice([2, 5, 9, 26, 27, 28, 40, 77]), 900 + rng.randrange(60)) for i in range(n)]
    voices = [rng.randrange(11) for _ in range(n)]
    lens = [rng.choice([1, 2, 3, 7, 19, 33, 48]) for _ in range(n)]
    arrive = sorted(rng.randrange(50) for _ in range(n))
    abort_at = {i: arrive[i] + rng.randrange(0, 12) for i in rng.sample(range(n), 8)}
    slots = rng.choice([2, 3, 6, 8])
    pool = slots * 2 * 3 - rng.choice([0, 3])
    eng = T3Engine(cfg, sd, max_model_len=192, max_num_seqs=slots, cfg_scale=3.0, kv_pages=max(pool, 6),
                   enforce_eager=rng.random() < 0.3, scheduling_policy=rng.choice(["fcfs", "longest_first", "longest_prompt_first"]),
                   max_num_batched_tokens=rng.choice([None, None, 24, 40, 64]))
    got, rid_of, aborted = {}, {}, set()
    step = nxt = 0
    while nxt < n or eng.has_unfinished():
        while nxt < n and arrive[nxt] <= step:
            rid_of[nxt] = eng.add_request(texts[nxt], conds[voices[nxt]], SamplingParams(temperature=0.0, repetition_penalty=2.0, max_tokens=lens[nxt], stop_token_ids=[STOP]))
            got[rid_of[nxt]] = []
            nxt += 1
        for i, at in abort_at.items():
            if at == step and i in rid_of and rid_of[i] not in aborted and len(got[rid_of[i]]) < lens[i]:
                try:
                    eng.abort(rid_of[i]); aborted.add(rid_of[i])
                except Exception:
                    pass                      # already finished
        for rid, tok, fin in eng.step():
            got[rid].append(tok)
        step += 1
        assert step < 20000
    for i in range(n):
        if rid_of[i] in aborted: continue
        if got[rid_of[i]] != solo_ids(texts[i], voices[i], lens[i]):
            bad += 1; print("MISMATCH seed", seed, "req", i, flush=True)
    assert eng.stats().free_pages == max(pool, 6), (seed, eng.stats().free_pages)
    eng.shutdown()
    print("seed", seed, "ok", flush=True)
print("bad", bad)
sys.exit(1 if bad else 0)
