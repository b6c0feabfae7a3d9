import os, sys, time, torch
sys.path.insert(0, "/root/repo")
from chatterbox_vllm_amd import cfm_weights as CW
from chatterbox_vllm_amd.cfm_decoder import CausalConditionalCFM
m = CausalConditionalCFM(CW.synthetic_state_dict(), max_frames=1000)
for L in (250, 500, 1000):
    inp = {k: v.cuda() for k, v in CW.synthetic_inputs(L, L // 3).items()}
    ts = CW.t_span(10)
    for _ in range(2):
        m.solve_euler(inp["noise"], ts, inp["mu"], inp["spks"], inp["cond"])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = m.solve_euler(inp["noise"], ts, inp["mu"], inp["spks"], inp["cond"])
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("frames %d: call returned after %.2f ms, device done after %.2f ms" % (L, 1e3 * (t1 - t0), 1e3 * (t2 - t0)))
