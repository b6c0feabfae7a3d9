#!/usr/bin/env python3
"""How much slower is a decode step while a prefill job shares the chip?  Runs the benchmark job and averages the
period of the step calls (each call waits for one decode step) by whether a prefill job was outstanding."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from chatterbox_vllm_amd.config import T3ModelConfig
from chatterbox_vllm_amd import weights as W
from chatterbox_vllm_amd.engine import T3Engine, SamplingParams


def main():
    cfg = T3ModelConfig.english()
    sd = W.synthetic_state_dict(cfg, seed=1234)
    conds = [W.synthetic_cond_emb(0)]
    job = bench.build_job(seed=7)
    eng = T3Engine(cfg, sd, max_model_len=1000, max_num_seqs=32, cfg_scale=3.0, scheduling_policy="longest_first")
    for rep in range(2):
        for text, out_len, voice in job:
            eng.add_request(text, conds[0], SamplingParams(temperature=0.0, repetition_penalty=2.0, max_tokens=out_len))
        rec = []
        t_prev = time.perf_counter()
        while eng.has_unfinished():
            eng.step()
            t = time.perf_counter()
            s = eng.stats()
            rec.append((t - t_prev, s.prefill_outstanding, s.decode_rows))
            t_prev = t
    full = [r for r in rec if r[2] >= 60]
    for flag in (0, 1):
        sel = [r[0] for r in full if (r[1] > 0) == bool(flag)]
        print(f"full-batch calls with{'' if flag else 'out'} a prefill job outstanding: {len(sel)}  mean period "
              f"{1e3 * sum(sel) / max(1, len(sel)):.3f} ms")
    print(f"all calls {len(rec)}  total {sum(r[0] for r in rec):.3f} s")


if __name__ == "__main__":
    main()
