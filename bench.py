#!/usr/bin/env python3
"""Benchmark of the T3 speech-token decode engine (the metric of BASELINE.json).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

`python bench.py --gpus N` with N > 1 from a plain shell starts its own N ranks (one child `torch.distributed.run`,
before this process has made any GPU call) and relays rank 0's line.

One STEP = one pass of the whole hot path over one batch of synthetic requests: the
154-chunk job of the reference's benchmark.py (text-length histogram of
docs/benchmark-text-1.txt, tests/golden/ref_text_fixture.json) is queued on one engine with 32
requests decoded concurrently (the queue is admitted longest PROMPT first, a ranking by what a caller knows;
--policy fcfs for arrival order, --policy longest_first for the ranking that also reads each request's max_tokens),
CFG-paired (cond + uncond stream, scale 3.0), greedy, repetition
penalty 2.0, prompt + output <= max_model_len 1000.  Random weights never emit the stop token
reliably, so every request runs to a fixed output length proportional to its text length with
the README-implied mean of 394 tokens.  Each rank runs the same job on its own GPU (weak
scaling: requests are independent; the only collective is the final all-gather of ids); value =
tokens of all ranks / max-over-ranks time.  --scaling strong deals ONE job (and the slots) over the ranks instead.

Other configurations of the same engine:
  --workload single         one request, CFG off (one KV stream), 152 text ids, decoded to position 1000
  --workload multilingual   multilingual vocabulary (2454), fr + zh chunk lengths of the reference's
                            benchmark-multilingual-{fr,zh}.py mixed in one queue, 4 voices; use --concurrency 64
  --quantization int8       W8A16 backbone weights; with --concurrency 128 --repeat-job 2 the int8 / batch-128 case
On one GPU the default command also runs them as short extra legs (one timed pass each) and reports them under
"other_configs", each with its own roofline object; so does the headline job under first-come-first-served admission
("fcfs_value"), ranked by prompt + max_tokens ("oracle_length_value": max_tokens is each request's exact output length
here, so this order knows what no caller knows) and with the uncond row that gets no attention
("uncond_no_attention_value").

Besides the throughput line this prints, in the same JSON object:
  roofline      the dominant kernel's algorithmic bytes / its HIP-event time, measured in a
                separate pass over the same job with every launch bracketed by events
  cpu_baseline  the CPU oracle (a port of the same arithmetic) timed on a bounded slice of the job
"""
import argparse
import hashlib
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from chatterbox_vllm_amd.config import T3ModelConfig, SPEECH_TOKEN_OFFSET  # noqa: E402,F401
from chatterbox_vllm_amd import weights as W  # noqa: E402

METRIC = "T3 speech-tokens/sec/GPU at batch=32, max_len=1000; 1/2/4/8-GPU scaling"
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak
MAX_MODEL_LEN = 1000
CONCURRENCY = 32
CFG_SCALE = 3.0
MEAN_OUT = 394                 # README-implied mean speech tokens per chunk (4577.88 tok/s * 13.25 s / 154)


def _fixture():
    with open(os.path.join(ROOT, "tests", "golden", "ref_text_fixture.json")) as f:
        return json.load(f)


def build_job(seed: int, n_requests: int = 0, workload: str = "benchmark", repeat: int = 1):
    """(text_ids, out_len, voice) per request.  Output length ~ text length, mean MEAN_OUT at the English
    job's mean text length, capped by max_model_len."""
    fx = _fixture()
    en_lens = fx["benchmark_text_1"]["text_id_counts"]
    mean_t = sum(en_lens) / len(en_lens)
    g = torch.Generator().manual_seed(seed)
    if workload == "single":
        spec = [(152, 255, 704, 0)]                        # (text ids, first id, vocab, voice)
    elif workload == "multilingual":
        fr = [(T, 634, 2454) for T in fx["benchmark_text_fr_1"]["text_id_counts"]]     # [fr] tag first (mtl_tokenizer.py:370)
        zh = [(T, 725, 2454) for T in fx["benchmark_text_zh_1"]["text_id_counts"]]     # [zh]
        mixed = [x for pair in zip(fr * 2, zh) for x in pair] + zh[len(fr) * 2:]
        mixed = (mixed * (128 // len(mixed) + 1))[:128]
        spec = [(T, first, vocab, i % 4) for i, (T, first, vocab) in enumerate(mixed)]
    else:
        spec = [(T, 255, 704, 0) for T in en_lens]
    spec = spec * repeat
    if n_requests:
        spec = spec[:n_requests]
    job = []
    for T, first, vocab, voice in spec:
        T = max(T, 3)
        body = torch.randint(3, vocab, (T - 2,), generator=g).tolist()
        text = [first] + body + [0]                     # [START] or language tag ... [STOP]
        P = 34 + T + 1
        out = MAX_MODEL_LEN - P if workload == "single" else int(round(MEAN_OUT * T / mean_t))
        out = max(16, min(out, MAX_MODEL_LEN - P))
        job.append((text, out, voice))
    return job


def run_job(eng, job, conds, stats=None):
    """Queue the whole job and run the engine dry.  Returns (#tokens, ids per request)."""
    from chatterbox_vllm_amd.engine import SamplingParams
    rids = []
    for text, out_len, voice in job:
        sp = SamplingParams(temperature=0.0, repetition_penalty=2.0, max_tokens=out_len, stop_token_ids=None)
        rids.append(eng.add_request(text, conds[voice], sp))
    ids = {r: [] for r in rids}
    n = 0
    while eng.has_unfinished():
        for rid, tok, _fin in eng.step():
            ids[rid].append(tok)
            n += 1
        if stats is not None:
            s = eng.stats()
            if s.decode_requests:
                stats["decode_steps"] += 1
                stats["decode_rows"] += s.decode_rows
                if "decode_requests" in stats:
                    stats["decode_requests"] += s.decode_requests
                stats["decode_ctx_tokens"] += s.decode_ctx_tokens
    return n, [ids[r] for r in rids]


def cpu_baseline(cfg, sd, job, conds, paired=True):
    """The oracle on a bounded slice: 8 requests of the job (the 8 shortest texts), CFG-paired,
    prefill + 96 greedy tokens each, batched decode steps (10-30 s of CPU work)."""
    from oracle.t3_oracle import T3Oracle
    torch.set_num_threads(max(1, min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16")))))
    o = T3Oracle(sd, n_layers=cfg.n_layers, cfg_scale=CFG_SCALE)
    sub = sorted(job, key=lambda t: len(t[0]))[:8]
    reqs = [(t, conds[v]) for t, _, v in sub]
    t0 = time.time()
    out = o.generate_batch(reqs, [96] * len(reqs), paired=paired)
    dt = time.time() - t0
    ntok = sum(len(x) for x in out)
    return {"value": round(ntok / dt, 3), "unit": "tokens/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{len(reqs)} shortest requests of the job (text ids {[len(t) for t, _, _ in sub]}), "
                      f"{'CFG-paired' if paired else 'single stream'}, "
                      f"prefill + 96 greedy tokens each, batched decode, {dt:.1f} s of CPU time"}, out, sub


def _kernel_source_sha():
    """sha256 (16 hex digits) over the kernel sources, so that a committed counter profile can be tied to the build."""
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "chatterbox_vllm_amd", "csrc")
    for name in sorted(os.listdir(csrc)):
        if name.endswith((".hip", ".h")):
            with open(os.path.join(csrc, name), "rb") as fh:
                h.update(fh.read())
    return h.hexdigest()[:16]


def _pmc_traffic(kernel_class):
    """HBM bytes per launch of a kernel class from the committed rocprofv3 --pmc passes over the default job
    (FETCH_SIZE doubled as gfx950 needs, plus WRITE_SIZE).  It is a committed profile, not a measurement of this run:
    returned only when the file says it was taken from the kernel sources this tree holds."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as fh:
            rec = json.load(fh)
    except (OSError, ValueError):
        return None
    if rec.get("kernel_source_sha16") not in (None, _kernel_source_sha()):
        return None
    return rec.get(kernel_class)


def stats_tokens(eng):
    """Tokens the decode steps of the event-timed pass sampled (requests per step summed), or 0 when not recorded."""
    return getattr(eng, "_timed_decode_tokens", 0)


def roofline_of(eng, job, conds, use_pmc):
    """One more pass of the job with every launch bracketed by HIP events bound to its dispatch; the roofline object of
    the dominant HBM-streaming kernel plus the per-class breakdown."""
    eng.set_kernel_timing(True)
    t1 = time.perf_counter()
    st = {"decode_steps": 0, "decode_rows": 0, "decode_ctx_tokens": 0, "decode_requests": 0}
    run_job(eng, job, conds, st)
    eng._timed_decode_tokens = st.get("decode_requests", 0)
    timed_pass = time.perf_counter() - t1
    kd, kp = eng.kernel_times("decode"), eng.kernel_times("prefill")
    eng.set_kernel_timing(False)
    tot = sum(v["ms"] for v in kd.values()) + sum(v["ms"] for v in kp.values())
    # the dominant kernel among those that stream from HBM (K/V or weights); the row-wise kernels (norm, embedding,
    # sampler) work out of L2 and have no HBM roofline
    streaming = {k: v for k, v in kd.items() if k.startswith(("attn_", "gemm_"))}
    name, dom = max(streaming.items(), key=lambda kv: kv[1]["ms"])
    per_launch_bytes = dom["bytes"] / dom["launches"]
    per_launch_s = dom["ms"] / dom["launches"] / 1000.0
    achieved = per_launch_bytes / per_launch_s / 1e9
    roof = {
        "bound": "hbm", "kernel": name, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
        "avg_launch_us": round(per_launch_s * 1e6, 2),
        "algorithmic_bytes_per_launch": int(per_launch_bytes),
        "launches": dom["launches"], "share_of_gpu_time": round(dom["ms"] / tot, 3),
    }
    pmc = _pmc_traffic(name) if use_pmc else None
    if pmc is not None:
        roof["traffic"] = pmc["bytes_per_launch"]
        roof["traffic_source"] = "committed profile (profiles/pmc_traffic.json): " + pmc["source"]
    breakdown = {
        ph: {k: {"ms": round(v["ms"], 2), "launches": v["launches"],
                 "GBps": round(v["bytes"] / max(v["ms"], 1e-9) / 1e6, 1)} for k, v in d.items()}
        for ph, d in (("decode", kd), ("prefill", kp))}
    breakdown["event_timed_pass_s"] = round(timed_pass, 3)
    # the decode step as a whole against the HBM roofline: algorithmic bytes of all its launches (weights once per step,
    # K/V of every stream, row gathers) over the summed launch durations; tokens = sampler rows of the decode steps
    d_ms = sum(v["ms"] for v in kd.values())
    d_steps = max(1, kd.get("sample", {}).get("launches", 0))
    d_bytes = sum(v["bytes"] for v in kd.values())
    d_tokens = stats_tokens(eng)
    breakdown["decode_step"] = {
        "steps": d_steps, "launch_ms_per_step": round(d_ms / d_steps, 4),
        "algorithmic_MB_per_step": round(d_bytes / d_steps / 1e6, 1),
        "achieved_GBps": round(d_bytes / max(d_ms, 1e-9) / 1e6, 1),
        "frac_of_hbm_peak": round(d_bytes / max(d_ms, 1e-9) / 1e6 / HBM_PEAK_GBS, 4),
        "decode_only_tokens_per_s": round(d_tokens / max(d_ms, 1e-9) * 1e3, 1) if d_tokens else None,
    }
    return roof, breakdown


ADMISSION_TEXT = {
    "fcfs": "fcfs: arrival order (vLLM's policy)",
    "longest_prompt_first": "longest_prompt_first: the whole queue is known up front (one generate() call) and ranked by "
                            "prompt length alone; the scheduler never reads max_tokens or the output length",
    "longest_first": "longest_first: ranked by prompt + max_tokens, and this job sets max_tokens to each request's exact "
                     "output length: knowledge a caller does not have"}


def workload_text(workload, conc):
    return {
        "benchmark": "benchmark.py 154-chunk job (docs/benchmark-text-1.txt length histogram), "
                     f"{conc} concurrent requests, CFG-paired scale 3.0, greedy, max_model_len 1000, "
                     f"output length ~ text length with mean {MEAN_OUT}; per GPU",
        "single": "one request, CFG off (single KV stream), 152 text ids, greedy, decoded to position 1000",
        "multilingual": "multilingual vocabulary 2454, fr + zh chunk lengths of benchmark-multilingual-{fr,zh}.py "
                        f"mixed in one queue of 128 requests, 4 voices, {conc} concurrent requests, CFG-paired "
                        "scale 3.0, greedy, max_model_len 1000; per GPU"}[workload]


def make_engine(cfg, sd, conc, paired, device, args, quantization, policy, uncond_attends=True):
    from chatterbox_vllm_amd.engine import T3Engine
    return T3Engine(cfg, sd, max_model_len=MAX_MODEL_LEN, max_num_seqs=conc, cfg_scale=CFG_SCALE,
                    cfg_paired=paired, device=device, max_prefill_rows=args.max_prefill_rows,
                    quantization=None if quantization == "none" else quantization,
                    scheduling_policy=policy, uncond_attends=uncond_attends,
                    prefix_sharing=not args.no_prefix_sharing, max_num_batched_tokens=args.max_num_batched_tokens or None)


FP32_MFMA_PEAK_TFLOPS = 157.3     # v_mfma_f32_32x32x2_f32, 64 FLOP per clock and SIMD, 1024 SIMDs at 2.4 GHz


def vocoder_leg(device, batch=8, frames=500, iters=5):
    """The token -> waveform side's last stage: HiFT vocoder decode() (mel + source -> 24 kHz waveform, fp32) for `batch`
    utterances of `frames` mel frames (10 s each at 500), inputs resident in HBM.  Its roofline is the fp32 matrix
    pipe: algorithmic FLOP of the convolutions (hift_weights.decode_flop) over the HIP-event time of the whole call on
    the call's stream, so the small kernels around the convolutions count against it.  The CPU figure is the oracle
    (torch fp32, the reference's own arithmetic) on the same batch, once."""
    from chatterbox_vllm_amd import hift_weights as HW
    from chatterbox_vllm_amd.hift import HiFTGenerator
    sd = HW.synthetic_state_dict()
    gen = HiFTGenerator(sd, max_batch=batch, max_frames=frames, device=device)
    mel = HW.synthetic_mel(frames, seed=3, batch=batch).to(f"cuda:{device}")
    src = gen.source(gen.f0_predictor(mel), seed=1)
    gen.decode(mel, src)
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for _ in range(iters):
        gen.decode(mel, src)
    torch.cuda.synchronize(device)
    wall = (time.perf_counter() - t0) / iters
    gen.set_timing(True)
    ev = []
    for _ in range(iters):
        gen.decode(mel, src)
        ev.append(gen.last_timing())
    gen.set_timing(False)
    ev_ms = sum(e[5] for e in ev) / len(ev)
    t0 = time.perf_counter()
    for _ in range(iters):
        gen.inference(mel, seed=2)
    torch.cuda.synchronize(device)
    wall_inf = (time.perf_counter() - t0) / iters
    gen.shutdown()
    pre, stages, post = HW.decode_flop(frames)
    flop = batch * (pre + sum(stages) + post)
    achieved = flop / (ev_ms * 1e-3) / 1e12
    # HBM bytes per call from the committed counter passes, quoted only for the kernel sources they were taken on
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", "hift_pmc_traffic.json")) as fh:
            prof = json.load(fh)
        h = hashlib.sha256()
        d = os.path.join(ROOT, "chatterbox_vllm_amd", "csrc_s3gen")
        for name in ("hift_engine.hip", "hift_kernels.hip"):     # the vocoder's kernels and their launch logic
            with open(os.path.join(d, name), "rb") as fh:
                h.update(fh.read())
        if prof.get("kernel_source_sha16") == h.hexdigest()[:16] and (batch, frames) == (8, 500):
            traffic = prof["bytes_per_call"]
    except (OSError, ValueError, KeyError):
        pass
    rec = {"workload": f"S3Gen HiFT vocoder, decode() of {batch} utterances x {frames} mel frames ({frames * 0.02:.0f} s each), seeded synthetic weights",
           "metric": "audio seconds per second", "value": round(batch * frames * 0.02 / wall, 1), "unit": "audio s/s",
           "ms_per_step": round(wall * 1e3, 3), "inference_value": round(batch * frames * 0.02 / wall_inf, 1), "dtype": "f32",
           "roofline": {"bound": "mfma", "kernel": "conv1d_kernel (74 launches of a call)", "achieved": round(achieved, 2),
                        "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / FP32_MFMA_PEAK_TFLOPS, 4),
                        "traffic": traffic, "traffic_unit": "HBM bytes per call (committed rocprofv3 --pmc passes)",
                        "event_ms_per_call": round(ev_ms, 3), "algorithmic_gflop_per_call": round(flop / 1e9, 1),
                        "stage_ms": [round(sum(e[i] for e in ev) / len(ev), 3) for i in range(5)]}}
    from oracle import hift_oracle as HO
    w = HO.fold_weight_norm(sd)
    mel_c, src_c = mel.cpu(), src.cpu()
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    t0 = time.perf_counter()
    HO.decode(w, mel_c, src_c)
    dt = time.perf_counter() - t0
    rec["cpu_baseline"] = {"value": round(batch * frames * 0.02 / dt, 2), "unit": "audio s/s", "cores": torch.get_num_threads(),
                           "kind": "port", "sample": f"one decode() of the same {batch} x {frames} frames, torch fp32, {dt:.1f} s"}
    return rec


def flow_decoder_leg(device, frames=1000, n_timesteps=10, iters=3, per_call=8):
    """The token -> waveform side's middle stage: the flow-matching decoder (mu -> mel, fp32), one utterance of `frames`
    mel frames (20 s at 1000, prompt included), `n_timesteps` Euler steps of two estimator rows each, inputs resident in
    HBM.  Its roofline is the fp32 matrix pipe: algorithmic FLOP of the estimator's matrix products
    (cfm_weights.estimator_flop) over the HIP-event time of the whole solve, so layer norms and launch gaps count against
    it.  `per_call` utterances of the same length through shared launches (s3cfm_solve_batch) are timed beside it.  The CPU
    figure is the restatement (torch fp32) on the same utterance, once."""
    from chatterbox_vllm_amd import cfm_weights as CW
    from chatterbox_vllm_amd.cfm_decoder import CausalConditionalCFM
    sd = CW.synthetic_state_dict()
    m = CausalConditionalCFM(sd, max_frames=frames, device=device, max_batch=per_call)
    dev = f"cuda:{device}"
    inp = {k: v.to(dev) for k, v in CW.synthetic_inputs(frames, frames // 3).items()}
    ts = CW.t_span(n_timesteps)
    m.solve_euler(inp["noise"], ts, inp["mu"], inp["spks"], inp["cond"])
    torch.cuda.synchronize(device)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(iters):
        m.solve_euler(inp["noise"], ts, inp["mu"], inp["spks"], inp["cond"])
    e1.record()
    torch.cuda.synchronize(device)
    wall = (time.perf_counter() - t0) / iters
    ev_ms = e0.elapsed_time(e1) / iters
    rep8 = {k: torch.cat([v] * per_call) for k, v in inp.items()}
    m.solve_euler_batch(rep8["noise"], ts, rep8["mu"], rep8["spks"], rep8["cond"], [frames] * per_call)
    e0.record()
    m.solve_euler_batch(rep8["noise"], ts, rep8["mu"], rep8["spks"], rep8["cond"], [frames] * per_call)
    e1.record()
    torch.cuda.synchronize(device)
    ev8_ms = e0.elapsed_time(e1)
    # the same solve with the matrix launches on split-bf16 operands (s3cfm_set_split_bf16: 1 = attention, 2 = linear layers and
    # convolutions too): labelled extras, the fp32 matrix instruction stays the leg's value; each one's distance from the fp32
    # form's result is reported with it
    ref32 = m.solve_euler(inp["noise"], ts, inp["mu"], inp["spks"], inp["cond"])
    split_runs = []
    for mode, label in ((1, "attention"), (2, "attention, linear layers and convolutions")):
        m.set_split_bf16(mode)
        got_split = m.solve_euler(inp["noise"], ts, inp["mu"], inp["spks"], inp["cond"])
        torch.cuda.synchronize(device)
        e0.record()
        for _ in range(iters):
            m.solve_euler(inp["noise"], ts, inp["mu"], inp["spks"], inp["cond"])
        e1.record()
        torch.cuda.synchronize(device)
        split_runs.append((label, e0.elapsed_time(e1) / iters, ((got_split - ref32).abs().max() / ref32.abs().max()).item()))
    m.shutdown()
    flop = CW.estimator_flop(frames) * n_timesteps
    achieved = flop / (ev_ms * 1e-3) / 1e12
    rec = {"workload": f"S3Gen flow-matching decoder, one utterance of {frames} mel frames ({frames * 0.02:.0f} s), {n_timesteps} Euler steps "
                       "x 2 guidance rows, seeded synthetic weights",
           "metric": "audio seconds per second", "value": round(frames * 0.02 / wall, 1), "unit": "audio s/s",
           "ms_per_step": round(wall * 1e3, 3), "dtype": "f32", "parity": "restatement pinned by outputs of the reference's decoder.py / flow_matching.py (tests/golden/cfm_fixture.npz); the inside of three diffusers classes unpinned",
           "roofline": {"bound": "mfma", "kernel": "lin_kernel + attn_kernel (about 3 700 launches of a solve)", "achieved": round(achieved, 2),
                        "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / FP32_MFMA_PEAK_TFLOPS, 4),
                        "traffic": None, "event_ms_per_call": round(ev_ms, 3), "algorithmic_gflop_per_call": round(flop / 1e9, 1)}}
    rec["several_per_call"] = {"utterances_per_call": per_call, "value": round(per_call * frames * 0.02 / (ev8_ms * 1e-3), 1),
                               "event_ms_per_call": round(ev8_ms, 3), "achieved": round(per_call * flop / (ev8_ms * 1e-3) / 1e12, 2),
                               "frac": round(per_call * flop / (ev8_ms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4)}
    rec["split_bf16"] = {"note": "fp32 operands as three bf16 values each, six partial products per product on the bf16 matrix instruction, "
                                 "fp32 accumulation; held to the fp32 form's tolerances against fp64 (tests/test_cfm_gpu.py runs every "
                                 "decoder test in all forms); off by default",
                         "forms": [{"launches": label, "value": round(frames * 0.02 / (ms * 1e-3), 1), "event_ms_per_call": round(ms, 3),
                                    "fp32_equivalent_tflops": round(flop / (ms * 1e-3) / 1e12, 1),
                                    "max_difference_from_the_fp32_form": float(f"{diff:.2e}")} for label, ms, diff in split_runs]}
    from oracle import cfm_oracle as CO
    small = frames
    inp_c = CW.synthetic_inputs(small, small // 3)
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    t0 = time.perf_counter()
    CO.solve_euler(sd, inp_c["noise"], ts, inp_c["mu"], inp_c["spks"], inp_c["cond"])
    dt = time.perf_counter() - t0
    rec["cpu_baseline"] = {"value": round(small * 0.02 / dt, 2), "unit": "audio s/s", "cores": torch.get_num_threads(), "kind": "port",
                           "sample": f"one solve of a {small}-frame utterance ({n_timesteps} steps), torch fp32, {dt:.1f} s"}
    return rec


def token_to_waveform_leg(device, prompt_tokens=250, new_tokens=250, iters=3):
    """The whole token -> waveform chain as the reference's ``S3Token2Wav.inference`` runs it per utterance
    (s3gen.py:296-317): `prompt_tokens` of voice reference (10 s) + `new_tokens` speech tokens (10 s of new audio) ->
    flow encoder -> flow-matching decoder (10 steps) -> vocoder -> trim fade, seeded synthetic weights, one utterance at a
    time.  `value` counts the generated audio only (the prompt's frames are computed and dropped, as in the reference)."""
    from chatterbox_vllm_amd import cfm_weights as CW, flow_weights as FW, hift_weights as HW
    from chatterbox_vllm_amd.s3gen import S3Token2Wav
    flow_sd = dict(FW.synthetic_state_dict())
    flow_sd.update(CW.synthetic_state_dict())
    s3 = S3Token2Wav(flow_sd, HW.synthetic_state_dict(), max_tokens=prompt_tokens + new_tokens, device=device)
    g = torch.Generator().manual_seed(12)
    ref = {"prompt_token": FW.synthetic_tokens(prompt_tokens, seed=3).view(1, -1),
           "prompt_feat": (2.0 * torch.randn(1, 2 * prompt_tokens, 80, generator=g) - 4.0).to(f"cuda:{device}"),
           "embedding": torch.randn(1, 192, generator=g).to(f"cuda:{device}")}
    new = FW.synthetic_tokens(new_tokens, seed=4).to(f"cuda:{device}")
    wav, _ = s3.inference(new, ref)
    torch.cuda.synchronize(device)
    stage = {}
    for name, fn in (("flow_encoder", lambda: s3.encoder.decoder_inputs(new.view(1, -1), ref["prompt_token"], ref["prompt_feat"], ref["embedding"])),
                     ("flow_decoder_and_encoder", lambda: s3.flow_inference(new, ref)),
                     ("whole", lambda: s3.inference(new, ref))):
        fn()
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        for _ in range(iters):
            fn()
        torch.cuda.synchronize(device)
        stage[name] = (time.perf_counter() - t0) / iters
    s3.shutdown()
    audio_s = new_tokens / 25.0
    return {"workload": f"S3Gen token -> waveform, one utterance: {prompt_tokens} prompt tokens + {new_tokens} new tokens "
                        f"({2 * (prompt_tokens + new_tokens)} mel frames through the flow stage, {audio_s:.0f} s of new audio), 10 Euler steps",
            "metric": "generated audio seconds per second", "value": round(audio_s / stage["whole"], 1), "unit": "audio s/s",
            "ms_per_step": round(stage["whole"] * 1e3, 2), "dtype": "f32", "finite": bool(torch.isfinite(wav).all()),
            "stage_ms": {"flow_encoder": round(stage["flow_encoder"] * 1e3, 2),
                         "flow_decoder": round((stage["flow_decoder_and_encoder"] - stage["flow_encoder"]) * 1e3, 2),
                         "vocoder_and_trim": round((stage["whole"] - stage["flow_decoder_and_encoder"]) * 1e3, 2)},
            "parity": "encoder, flow decoder and vocoder pinned by reference outputs (flow decoder: the inside of three diffusers classes unpinned)"}


def s3gen_job_leg(device, lengths, prompt_tokens=250, variants=((2, 4), (1, 1), (1, 8), (3, 4), (2, 4, 2))):
    """The second half of the reference's benchmark.py for the same job: every chunk's speech tokens through the token ->
    waveform chain with a 10 s voice prompt.  The token counts are the ones the T3 leg generated for the job's chunks; the
    token values and all weights are seeded synthetic.  `variants` are (copies of the chain in flight, utterances per
    call), all through S3Token2Wav.render_many: several chunks of similar length per call share the flow-matching stage's
    launches, several copies (each with its own weights, workspace, stream and host thread) use the CUs that one copy's
    launches leave idle; chunks are handed out longest first.  `value` is the FIRST variant, (2, 4): the schedule the
    facade builds by default (ChatterboxTTS.from_local / from_synthetic).  (1, 1) is one utterance at a time, as the
    reference runs it (tts.py:428-557).  A third entry 1 or 2 repeats a schedule with the flow-matching decoder's attention
    (1) or all its matrix launches (2) on split-bf16 operands (off by default; reported as `split_bf16`, never as `value`).  Reported beside the T3 time so that the whole text -> waveform job has a number
    (the reference's README reports the two times for its run, README.md:150-153)."""
    from chatterbox_vllm_amd import cfm_weights as CW, flow_weights as FW, hift_weights as HW
    from chatterbox_vllm_amd.s3gen import S3Token2Wav
    flow_sd = dict(FW.synthetic_state_dict())
    flow_sd.update(CW.synthetic_state_dict())
    hift_sd = HW.synthetic_state_dict()
    lengths = [n for n in lengths if n > 0]
    g = torch.Generator().manual_seed(12)
    dev = f"cuda:{device}"
    ref = {"prompt_token": FW.synthetic_tokens(prompt_tokens, seed=3).view(1, -1),
           "prompt_feat": (2.0 * torch.randn(1, 2 * prompt_tokens, 80, generator=g) - 4.0).to(dev),
           "embedding": torch.randn(1, 192, generator=g).to(dev)}
    toks = [torch.randint(0, 6561, (n,), generator=g).to(dev) for n in lengths]
    longest = sorted(range(len(toks)), key=lambda i: -lengths[i])
    runs = []
    for var in variants:
        nw, nb, split = var[0], var[1], (int(var[2]) if len(var) > 2 else 0)
        s3 = S3Token2Wav(flow_sd, hift_sd, max_tokens=prompt_tokens + max(lengths), device=device, max_batch=nb, chains=nw,
                         split_bf16=split)
        s3.render_many([toks[i] for i in longest[:2 * nw * nb]], ref)          # warm-up of every copy
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        wavs = s3.render_many(toks, ref)
        torch.cuda.synchronize(device)
        dt = time.perf_counter() - t0
        samples = sum(w.shape[1] for w in wavs)
        s3.shutdown()
        runs.append({"utterances_in_flight": nw, "utterances_per_call": nb, "seconds": round(dt, 3),
                     "value": round(samples / 24000 / dt, 1), "audio_seconds": round(samples / 24000, 1), "split": split})
    first = runs[0]
    rec = {"workload": f"S3Gen over the job's {len(lengths)} chunks ({sum(lengths)} speech tokens, {min(lengths)}-{max(lengths)} per chunk), "
                       f"{prompt_tokens}-token voice prompt, 10 Euler steps, seeded synthetic weights",
           "metric": "generated audio seconds per second", "value": first["value"], "unit": "audio s/s",
           "schedule": "the facade's default: two copies of the chain, four utterances per call, longest first",
           "utterances_in_flight": first["utterances_in_flight"], "utterances_per_call": first["utterances_per_call"],
           "seconds": first["seconds"], "audio_seconds": first["audio_seconds"], "dtype": "f32"}
    rec["variants"] = [{k: v for k, v in r.items() if k not in ("audio_seconds", "split")} for r in runs[1:] if not r["split"]]
    sp = next((r for r in runs if r["split"]), None)
    if sp:
        rec["split_bf16"] = {"launches": "attention" if sp["split"] == 1 else "attention, linear layers and convolutions",
                             "utterances_in_flight": sp["utterances_in_flight"], "utterances_per_call": sp["utterances_per_call"],
                             "seconds": sp["seconds"], "value": sp["value"]}
    rec["one_utterance_at_a_time_seconds"] = next((r["seconds"] for r in runs if (r["utterances_in_flight"], r["utterances_per_call"]) == (1, 1)
                                                   and not r["split"]), None)
    return rec


def extra_leg(args, device, sd_cache, workload, conc, quantization, repeat, label):
    """One short leg of another BASELINE configuration on this GPU: one warm-up pass, one timed pass, one event-timed
    pass.  Returns the record that goes under "other_configs"."""
    single = workload == "single"
    mtl = workload == "multilingual"
    key = "mtl" if mtl else "en"
    if key not in sd_cache:
        cfg = (T3ModelConfig.multilingual if mtl else T3ModelConfig.english)(n_layers=args.layers)
        sd_cache[key] = (cfg, W.synthetic_state_dict(cfg, seed=1234))
    cfg, sd = sd_cache[key]
    conds = [W.synthetic_cond_emb(v) for v in range(4)]
    job = build_job(seed=7, n_requests=args.requests, workload=workload, repeat=repeat)
    eng = make_engine(cfg, sd, conc, not single, device, args, quantization, args.policy)
    run_job(eng, job, conds)                                   # warm-up: graphs captured, caches warm
    stats = {"decode_steps": 0, "decode_rows": 0, "decode_ctx_tokens": 0}
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    n, _ = run_job(eng, job, conds, stats)
    torch.cuda.synchronize(device)
    dt = time.perf_counter() - t0
    rec = {"config": label, "workload": workload_text(workload, conc), "value": round(n / dt, 2), "unit": "tokens/s",
           "seconds": round(dt, 3), "requests": len(job), "concurrency": conc,
           "weights": "int8 (W8A16)" if quantization == "int8" else "bf16",
           "decode": {"steps": stats["decode_steps"],
                      "mean_rows": round(stats["decode_rows"] / max(1, stats["decode_steps"]), 2),
                      "mean_ctx": round(stats["decode_ctx_tokens"] / max(1, stats["decode_rows"]), 1)}}
    if not args.no_roofline:
        rec["roofline"], _ = roofline_of(eng, job, conds, use_pmc=False)
    eng.shutdown()
    return rec


def visible_gpu_count():
    """GPUs this process may use, counted without any HIP or torch.cuda call (the parent of the ranks must not
    initialise the GPU): the compute nodes of /sys/class/kfd/kfd/topology/nodes that have SIMDs, cut down by
    ROCR_VISIBLE_DEVICES / HIP_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES when one of them lists devices.  None when the
    topology cannot be read: the ranks then find out themselves."""
    nodes = "/sys/class/kfd/kfd/topology/nodes"
    try:
        n = 0
        for d in os.listdir(nodes):
            with open(os.path.join(nodes, d, "properties")) as fh:
                props = dict(ln.split()[:2] for ln in fh if len(ln.split()) >= 2)
            n += int(props.get("simd_count", "0")) > 0
    except (OSError, ValueError):
        return None
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip()]))
    return n


def launch_ranks(args) -> int:
    """`bench.py --gpus N` from a plain shell: start N ranks as ONE child process (torch.distributed.run) and pass its
    exit code on.  Called before this process has made any GPU call; this process only waits."""
    import socket
    import subprocess
    if args.backend == "nccl" and not args.rehearse:
        n = visible_gpu_count()
        if n is not None and n < args.gpus:
            print(f"bench.py: --gpus {args.gpus} needs {args.gpus} GPUs, {n} visible "
                  "(--backend gloo lets ranks share a GPU to rehearse the multi-rank path)", file=sys.stderr)
            return 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


class _RehearsalEngine:
    """--rehearse: no GPU and no library; stands in for the engine so that the launcher, the rendezvous, the broadcast,
    the gather and the reduction can be run on a CPU-only machine.  Its line says "rehearsal" and is not a measurement."""

    def __init__(self):
        self._q = []
        self._n = 0

    def add_request(self, text, cond, sp):
        self._n += 1
        self._q.append([self._n, min(4, sp.max_tokens), len(text)])
        return self._n

    def has_unfinished(self):
        return bool(self._q)

    def step(self):
        out = []
        for r in self._q:
            r[1] -= 1
            out.append((r[0], 1000 + (7 * r[2] + r[1]) % 6000, "length" if r[1] == 0 else None))
        self._q = [r for r in self._q if r[1] > 0]
        return out

    def stats(self):
        class S:
            decode_requests = 0
        return S()

    def shutdown(self):
        pass


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--requests", type=int, default=0, help="use only the first N requests of the job (debug)")
    ap.add_argument("--layers", type=int, default=30, help="debug only: anything but 30 is not the benchmark")
    ap.add_argument("--workload", choices=["benchmark", "single", "multilingual"], default="benchmark")
    ap.add_argument("--concurrency", type=int, default=0, help="requests decoded together per job (default 32; single: 1)")
    ap.add_argument("--quantization", choices=["none", "int8"], default="none")
    ap.add_argument("--repeat-job", type=int, default=1, help="queue the request list this many times per step")
    ap.add_argument("--max-num-batched-tokens", type=int, default=0,
                    help="vLLM's token budget for prompt steps: a prefill job carries at most this many positions per stream and "
                         "cuts the prompt that does not fit (0 = whole prompts, the default; tokens do not depend on it)")
    ap.add_argument("--policy", choices=["fcfs", "longest_prompt_first", "longest_first"], default="longest_prompt_first",
                    help="admission order of the queue.  longest_prompt_first (default) ranks the waiting requests by "
                         "prompt length alone, which every caller knows; it keeps the 32 slots full until the queue is "
                         "nearly dry.  fcfs is vLLM's order.  longest_first also reads max_tokens, which this job sets "
                         "to each request's exact output length: a ranking with knowledge a caller does not have, "
                         "reported as an extra.  Token ids are the same under all three")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak: every rank runs the whole job with all the slots.  strong: ONE job is dealt round-robin "
                         "over the ranks and the slots are divided among them (config 4: 128 slots / N)")
    ap.add_argument("--uncond-no-attention", action="store_true",
                    help="engine mode cfg_paired=2: the uncond row has no attention and no KV (not the benchmark)")
    ap.add_argument("--max-prefill-rows", type=int, default=2048,
                    help="prompt positions (per stream) one prefill job may hold; vLLM's counterpart is max_num_batched_tokens")
    ap.add_argument("--no-prefix-sharing", action="store_true",
                    help="every stream prefills and keeps its own copy of the 34 conditioning positions (same ids)")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise the process group even with one rank (under torchrun --nproc-per-node 1): runs the "
                         "RCCL broadcast / all-reduce / all-gather code path on a one-GPU box")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="collective backend under torchrun; gloo (CPU tensors, ranks may share a GPU) rehearses the "
                         "multi-rank path on a box with fewer GPUs than ranks")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-extra-legs", action="store_true",
                    help="skip the short legs of the other configurations (FCFS, single stream, int8 x 128, multilingual x 64)")
    ap.add_argument("--rehearse", action="store_true",
                    help="no GPU, no library: a stand-in engine, to run the launcher and the collectives on a CPU-only "
                         "machine (gloo); the line it prints says so and is not a measurement")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.rehearse:
        args.backend = "gloo"
    # stdout carries the one JSON line and nothing else: whatever libraries print there while the job runs (RCCL's
    # version banner, for one) goes to stderr instead
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    dist = None
    if world > 1 or (args.force_dist and "RANK" in os.environ):
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "gloo":
            local_rank %= max(1, torch.cuda.device_count())
            dist.init_process_group("gloo")
        else:
            if local_rank >= torch.cuda.device_count():
                sys.exit(f"bench.py: rank {rank} needs GPU {local_rank}, {torch.cuda.device_count()} visible")
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    coll_dev = torch.device("cpu") if args.backend == "gloo" else torch.device("cuda", local_rank)

    single = args.workload == "single"
    paired = not single
    conc = args.concurrency or (1 if single else CONCURRENCY)
    if args.scaling == "strong":
        conc = max(1, conc // world)                       # the slots are divided over the ranks
    n_layers = 0 if args.rehearse else args.layers
    cfg = (T3ModelConfig.multilingual if args.workload == "multilingual" else T3ModelConfig.english)(n_layers=n_layers)
    sd = W.synthetic_state_dict(cfg, seed=1234)
    if dist is not None:
        # rank 0's checkpoint is the checkpoint: one RCCL broadcast of the weights over xGMI
        from chatterbox_vllm_amd.sharding import broadcast_state_dict
        sd = broadcast_state_dict(sd, src=0, device=coll_dev)
    conds = [W.synthetic_cond_emb(v) for v in range(4)]
    if args.scaling == "strong":
        from chatterbox_vllm_amd.sharding import shard_indices
        whole = build_job(seed=7, n_requests=args.requests, workload=args.workload, repeat=args.repeat_job)
        job = [whole[i] for i in shard_indices(len(whole), world, rank)]
    else:
        job = build_job(seed=7 + rank, n_requests=args.requests, workload=args.workload, repeat=args.repeat_job)
    if args.rehearse:
        eng = _RehearsalEngine()
    else:
        eng = make_engine(cfg, sd, conc, paired, local_rank, args, args.quantization, args.policy,
                          uncond_attends=not args.uncond_no_attention)

    def barrier():
        if dist is not None:
            dist.barrier()
        if not args.rehearse:
            torch.cuda.synchronize(local_rank)

    for _ in range(args.warmup):
        run_job(eng, job, conds)
    stats = {"decode_steps": 0, "decode_rows": 0, "decode_ctx_tokens": 0}
    barrier()
    t0 = time.perf_counter()
    tokens = 0
    last_ids = None
    for _ in range(args.steps):
        n, last_ids = run_job(eng, job, conds, stats)
        tokens += n
    barrier()
    dt = time.perf_counter() - t0
    ranks_seen = 1
    if dist is not None:
        t = torch.tensor([dt], device=coll_dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        tk = torch.tensor([tokens, 1], device=coll_dev, dtype=torch.int64)
        dist.all_reduce(tk, op=dist.ReduceOp.SUM)
        total_tokens, ranks_seen = int(tk[0].item()), int(tk[1].item())
        # the job's result travels to rank 0 the way a sharded generate() returns it
        from chatterbox_vllm_amd.sharding import gather_token_ids
        gathered = gather_token_ids(last_ids, device=coll_dev)
        n_jobs = torch.tensor([len(job)], device=coll_dev, dtype=torch.int64)
        dist.all_reduce(n_jobs, op=dist.ReduceOp.SUM)
        assert len(gathered) == int(n_jobs.item())
    else:
        total_tokens = tokens

    result = None
    if rank == 0:
        result = {
            "metric": METRIC, "value": round(total_tokens / dt, 2), "unit": "tokens/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1000 * dt / args.steps, 3),
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "bf16",
            "data": "rehearsal: stand-in engine, no GPU work, not a measurement" if args.rehearse else "synthetic",
            "ranks_seen": ranks_seen,
            "config": {"workload": workload_text(args.workload, conc),
                       "requests_per_gpu": len(job), "tokens_per_step_per_gpu": tokens // max(1, args.steps),
                       "concurrency": conc, "layers": cfg.n_layers, "weights": "int8 (W8A16)" if args.quantization == "int8" else "bf16",
                       "admission": ADMISSION_TEXT[args.policy], "uncond_stream": "no attention, no KV" if args.uncond_no_attention else "own KV",
                       "cond_prefix_kv": "per stream" if args.no_prefix_sharing else "shared per voice",
                       "parallelism": f"request-sharded x{world}"},
            "per_gpu_tokens_per_s": round(total_tokens / dt / world, 2),
            "decode": {"steps": stats["decode_steps"],
                       "mean_rows": round(stats["decode_rows"] / max(1, stats["decode_steps"]), 2),
                       "mean_ctx": round(stats["decode_ctx_tokens"] / max(1, stats["decode_rows"]), 1)},
        }

    # ---- roofline: per-launch event timing of every kernel class over one more pass of the job
    if rank == 0 and not args.no_roofline and not args.rehearse:
        default_job = (args.workload == "benchmark" and args.quantization == "none" and not args.requests
                       and not args.uncond_no_attention and args.layers == 30 and not args.concurrency)
        result["roofline"], result["kernel_breakdown"] = roofline_of(eng, job, conds, use_pmc=default_job)

    # ---- the same job under the admission order vLLM has (first come first served), driver-timed like the headline
    extras = rank == 0 and world == 1 and not args.no_extra_legs and not args.rehearse and args.workload == "benchmark" \
        and args.quantization == "none" and not args.uncond_no_attention
    if extras:
        for pol, key in (("fcfs", "fcfs_value"), ("longest_first", "oracle_length_value")):
            if pol == args.policy:
                continue
            eng.set_schedule_policy(pol)
            run_job(eng, job, conds)
            torch.cuda.synchronize(local_rank)
            t1 = time.perf_counter()
            n_f, ids_f = run_job(eng, job, conds)
            torch.cuda.synchronize(local_rank)
            result[key] = round(n_f / (time.perf_counter() - t1), 2)
            result[key.replace("value", "ids_equal")] = ids_f == last_ids
        result["oracle_length_note"] = ADMISSION_TEXT["longest_first"]
    eng.shutdown()

    if extras:
        # the uncond row without attention or KV (what a serving stack computes if it attends only the rows its
        # scheduler counted); then BASELINE configs 2, 4 (one GPU's view: all 128 slots here) and 5
        e2 = make_engine(cfg, sd, conc, paired, local_rank, args, "none", args.policy, uncond_attends=False)
        run_job(e2, job, conds)
        torch.cuda.synchronize(local_rank)
        t1 = time.perf_counter()
        n2, _ = run_job(e2, job, conds)
        torch.cuda.synchronize(local_rank)
        result["uncond_no_attention_value"] = round(n2 / (time.perf_counter() - t1), 2)
        e2.shutdown()
        cache = {"en": (cfg, sd)}
        result["other_configs"] = [
            extra_leg(args, local_rank, cache, "single", 1, "none", 1, "configs[1]: bf16, batch 1, CFG off"),
            extra_leg(args, local_rank, cache, "benchmark", 128, "int8", 2,
                      "configs[3] on one GPU: int8 weights, 128 slots, continuous admission from a 308-request queue"),
            extra_leg(args, local_rank, cache, "multilingual", 64, "none", 1,
                      "configs[4]: multilingual vocabulary, fr + zh mixed, 64 slots, 4 voices"),
        ]
        result["vocoder"] = vocoder_leg(local_rank)
        result["flow_decoder"] = flow_decoder_leg(local_rank)
        result["token_to_waveform"] = token_to_waveform_leg(local_rank)
        # the job's own chunks through the waveform stages: ids below the stop / offset rules' reach, so the counts are the job's
        result["s3gen_job"] = s3gen_job_leg(local_rank, [len(x) for x in last_ids])
        result["s3gen_job"]["t3_seconds_same_job"] = round(result["ms_per_step"] / 1e3, 3)

    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.rehearse:   # the CPU leg is timed on the one-GPU run only
        cb, cpu_ids, sub = cpu_baseline(cfg, sd, job, conds, paired=paired)
        result["cpu_baseline"] = cb
    sys.stdout.flush()
    os.dup2(json_fd, 1)
    if rank == 0:
        print(json.dumps(result), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
