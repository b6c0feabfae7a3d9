#!/usr/bin/env python3
"""Do a prefill chain and a decode chain overlap when they run on two HIP streams?  Two engines in one process
(each has its own stream): A only prefills (requests of one token), B only decodes (32 long requests).  Prints
the wall time of A alone, B alone and both driven from two threads at once."""
import os, sys, threading, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chatterbox_vllm_amd.config import T3ModelConfig
from chatterbox_vllm_amd import weights as W
from chatterbox_vllm_amd.engine import T3Engine, SamplingParams


def main():
    cfg = T3ModelConfig.english()
    sd = W.synthetic_state_dict(cfg, seed=1234)
    cond = W.synthetic_cond_emb(0)
    g = torch.Generator().manual_seed(3)
    mk = lambda T: [255] + torch.randint(3, 704, (T - 2,), generator=g).tolist() + [0]
    A = T3Engine(cfg, sd, max_model_len=1000, max_num_seqs=2, cfg_scale=3.0, max_prefill_rows=600)
    B = T3Engine(cfg, sd, max_model_len=1000, max_num_seqs=32, cfg_scale=3.0, scheduling_policy="longest_first")
    a_items = [mk(152) for _ in range(120)]
    b_items = [mk(150) for _ in range(32)]

    def run_a():
        for t in a_items:
            A.add_request(t, cond, SamplingParams(temperature=0.0, repetition_penalty=2.0, max_tokens=1))
        while A.has_unfinished():
            A.step()

    def run_b():
        for t in b_items:
            B.add_request(t, cond, SamplingParams(temperature=0.0, repetition_penalty=2.0, max_tokens=200))
        while B.has_unfinished():
            B.step()

    def timed(fns):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        th = [threading.Thread(target=f) for f in fns]
        for t in th: t.start()
        for t in th: t.join()
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    timed([run_a]); timed([run_b])
    ta, tb, tab = timed([run_a]), timed([run_b]), timed([run_a, run_b])
    print(f"prefill-only engine {ta*1e3:.1f} ms   decode-only engine {tb*1e3:.1f} ms   both at once {tab*1e3:.1f} ms "
          f"(sum {1e3*(ta+tb):.1f}, max {1e3*max(ta,tb):.1f})")


if __name__ == "__main__":
    main()
