"""GPU tests of the flow-matching decoder (include/s3cfm_hip.h) against oracle/cfm_oracle.py run in fp64 on the same seeded
weights and inputs, and against tests/golden/cfm_fixture.npz: outputs of the reference's own ConditionalDecoder.forward and
CausalConditionalCFM.forward on those weights (made by tests/golden/make_cfm_fixtures.py; the three classes the reference
takes from diffusers are labelled stand-ins there, so the inside of `Attention`, `GELU` and `LoRACompatibleLinear` is the one
part these tests cannot pin).

Tolerances are relative to the largest magnitude of the compared tensor: 2e-5 for one estimator call (about 70 layers of
fp32 arithmetic against fp64; 3e-5 against the reference's fp32 output, which carries its own rounding), 1e-4 for the
ten-step solve (the fp32 restatement itself sits 1e-5 from the fp64 one)."""
import os

import numpy as np
import pytest
import torch

from chatterbox_vllm_amd import cfm_weights as CW

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sd():
    return CW.synthetic_state_dict()


# every test of the decoder runs in all three arithmetic forms (fp32 matrix instruction; attention on split-bf16 operands;
# attention, linear layers and convolutions on them), to the same tolerances
_FORMS = dict(params=[False, True, 2], ids=["fp32", "split_bf16_attention", "split_bf16_everywhere"])


@pytest.fixture(scope="module", **_FORMS)
def cfm(sd, request):
    from chatterbox_vllm_amd.cfm_decoder import CausalConditionalCFM
    m = CausalConditionalCFM(sd, max_frames=700, generator=torch.Generator().manual_seed(3), split_bf16=request.param)
    yield m
    m.shutdown()


def _rel(a, b):
    return (a.double().cpu() - b).abs().max().item() / b.abs().max().item()


@pytest.mark.parametrize("L,t", [(50, 0.0), (131, 0.4122), (300, 0.95)])
def test_estimator_matches_the_restatement(cfm, sd, L, t):
    from oracle import cfm_oracle as O
    g = torch.Generator().manual_seed(L)
    x = torch.randn(2, 80, L, generator=g)
    mu = torch.randn(2, 80, L, generator=g)
    spks = 0.5 * torch.randn(2, 80, generator=g)
    cond = torch.randn(2, 80, L, generator=g)
    got = cfm.estimator(x, mu, t, spks, cond)
    w = {k: v.double() for k, v in sd.items()}
    # the time value the kernel sees is the fp32 one
    tt = torch.tensor([t, t], dtype=torch.float32).double()
    want = O.estimator(w, x.double(), mu.double(), tt, spks.double(), cond.double())
    assert _rel(got, want) < 2e-5


@pytest.fixture(scope="module")
def fx():
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "cfm_fixture.npz"))
    return {k: torch.from_numpy(z[k]) if z[k].ndim else z[k].item() for k in z.files}


@pytest.mark.parametrize("tag", ["b1", "b2"])
def test_estimator_matches_the_reference_module_tree(cfm, fx, tag):
    """s3cfm_estimator against the reference's own ConditionalDecoder.forward (decoder.py:233-317) on the seeded weights."""
    i = CW.synthetic_inputs(fx[f"{tag}_frames"], fx[f"{tag}_prompt"], seed=fx[f"{tag}_seed"])
    z = torch.zeros_like(i["mu"])
    got = cfm.estimator(torch.cat([i["noise"], i["noise"]]), torch.cat([i["mu"], z]), float(fx[f"{tag}_t"]),
                        torch.cat([i["spks"], torch.zeros(1, 80)]), torch.cat([i["cond"], z]))
    assert _rel(got, fx[f"{tag}_estimator_out"].double()) < 3e-5


def test_solve_matches_the_reference_solve(cfm, fx):
    """s3cfm_solve against the reference's CausalConditionalCFM.forward around its own estimator (flow_matching.py:203-228,
    82-133), 10 steps, 40 frames."""
    i = CW.synthetic_inputs(40, 12)
    got = cfm.solve_euler(i["noise"], CW.t_span(10), i["mu"], i["spks"], i["cond"])
    assert _rel(got, fx["b_forward_10"].double()) < 1e-4


def test_solve_matches_the_restatement(cfm, sd):
    from oracle import cfm_oracle as O
    L = 96
    inp = CW.synthetic_inputs(L, 30)
    ts = CW.t_span(10)
    got = cfm.solve_euler(inp["noise"], ts, inp["mu"], inp["spks"], inp["cond"])
    want = O.solve_euler(sd, inp["noise"], ts, inp["mu"], inp["spks"], inp["cond"], dtype=torch.float64)
    assert _rel(got, want) < 1e-4
    # forward() slices the fixed noise buffer and builds the same grid (flow_matching.py:223-228)
    got2, _ = cfm(mu=inp["mu"], mask=torch.ones(1, 1, L), spks=inp["spks"], cond=inp["cond"], n_timesteps=10)
    want2 = O.cfm_forward(sd, inp["mu"], inp["spks"], inp["cond"], cfm.rand_noise, dtype=torch.float64)
    assert _rel(got2, want2) < 1e-4


def test_one_step_is_the_guidance_formula(cfm):
    # flow_matching.py:125-127 with the estimator's own two rows
    L = 64
    inp = CW.synthetic_inputs(L, 10)
    ts = torch.tensor([0.0, 0.25])
    x1 = cfm.solve_euler(inp["noise"], ts, inp["mu"], inp["spks"], inp["cond"])
    z = inp["noise"].cuda()
    zero = torch.zeros_like(z)
    d = cfm.estimator(torch.cat([z, z]), torch.cat([inp["mu"].cuda(), zero]), 0.0, torch.cat([inp["spks"].cuda(), torch.zeros(1, 80).cuda()]),
                      torch.cat([inp["cond"].cuda(), zero]))
    want = z + 0.25 * ((1.0 + 0.7) * d[0:1] - 0.7 * d[1:2])
    assert torch.equal(x1, want)


def test_repeat_calls_and_shorter_utterance_after_longer(cfm):
    a = CW.synthetic_inputs(200, 40, seed=1)
    b = CW.synthetic_inputs(72, 16, seed=2)
    ts = CW.t_span(3)
    b1 = cfm.solve_euler(b["noise"], ts, b["mu"], b["spks"], b["cond"])
    cfm.solve_euler(a["noise"], ts, a["mu"], a["spks"], a["cond"])
    b2 = cfm.solve_euler(b["noise"], ts, b["mu"], b["spks"], b["cond"])
    assert torch.equal(b1, b2)


def test_limits(cfm):
    inp = CW.synthetic_inputs(701, 10)
    with pytest.raises(ValueError):
        cfm.solve_euler(inp["noise"], CW.t_span(2), inp["mu"], inp["spks"], inp["cond"])


@pytest.mark.parametrize("split", [False, True], ids=["fp32", "split_bf16"])
def test_token_to_waveform_chain_with_the_packaged_decoder(sd, split):
    """S3Token2Wav with its default decoder: tokens -> mu (FlowEncoder) -> mel (CausalConditionalCFM) -> waveform
    (HiFTGenerator).  The mel is held to the two restatements chained in fp64 (flow.py:262-291 with finalize=True); the
    waveform is the vocoder's own for that mel.  Both arithmetic forms of the decoder's attention, through the chain's own
    switch."""
    from chatterbox_vllm_amd import flow_weights as FW, hift_weights as HW
    from chatterbox_vllm_amd.s3gen import S3Token2Wav
    from oracle import cfm_oracle as O, flow_encoder_oracle as FO
    flow_sd = dict(FW.synthetic_state_dict())
    flow_sd.update(sd)
    torch.manual_seed(11)                                    # the decoder draws its fixed noise at construction
    s3 = S3Token2Wav(flow_sd, HW.synthetic_state_dict(), max_tokens=64, split_bf16=split)
    assert s3.decoder.split_bf16 == split
    n_prompt, n_new = 14, 30
    g = torch.Generator().manual_seed(8)
    ref = {"prompt_token": FW.synthetic_tokens(n_prompt, seed=3).view(1, -1),
           "prompt_feat": 2.0 * torch.randn(1, 2 * n_prompt, 80, generator=g) - 4.0,
           "embedding": torch.randn(1, 192, generator=g)}
    new = FW.synthetic_tokens(n_new, seed=4)
    mel = s3.flow_inference(new, ref, n_timesteps=10)
    assert mel.shape == (1, 80, 2 * n_new)
    w64 = {k: v.double() for k, v in flow_sd.items()}
    tokens = torch.cat([ref["prompt_token"].view(-1), new])
    mu = FO.token_to_mu(w64, tokens, dtype=torch.float64).t().unsqueeze(0)
    spks = FO.speaker_projection(w64, ref["embedding"], dtype=torch.float64).view(1, 80)
    cond = torch.zeros(1, 80, mu.shape[2], dtype=torch.float64)
    cond[:, :, :2 * n_prompt] = ref["prompt_feat"].double().transpose(1, 2)
    want = O.cfm_forward(flow_sd, mu, spks, cond, s3.decoder.rand_noise, dtype=torch.float64)[:, :, 2 * n_prompt:]
    assert _rel(mel, want) < 1e-4
    wav, _ = s3.inference(new, ref, seed=3)
    raw, _ = s3.hift_inference(mel, seed=3)
    assert wav.shape == (1, 2 * n_new * 480) and torch.equal(wav[0, 960:], raw[0, 960:]) and not wav[0, :480].any()
    s3.shutdown()


@pytest.mark.parametrize("split", [False, True], ids=["fp32", "split_bf16"])
@pytest.mark.parametrize("Cin,Cout,K,pad,L", [(256, 256, 1, 0, 1000), (320, 256, 3, 2, 77), (1024, 256, 1, 0, 333), (512, 256, 3, 2, 50),
                                               (256, 1024, 1, 0, 1), (256, 80, 1, 0, 97), (256, 1024, 1, 0, 4200), (256, 1024, 1, 0, 1000), (256, 1536, 1, 0, 1000), (1024, 256, 1, 0, 1100),
                                               (256, 512, 1, 0, 8200), (320, 512, 3, 2, 8199)])
def test_linear_kernel_against_fp64(Cin, Cout, K, pad, L, split):
    """Every shape class the estimator launches: the four workgroup forms (8-, 4- and 2-way split of the reduction, none:
    chosen by the number of tiles; 1100 frames x 256 channels takes one 32-channel block per wave), causal taps, ragged
    lengths, 80 output channels, GELU and the residual add.  The launches of 8200 frames take the form without a split of the
    reduction, whose split-bf16 variant shares a workgroup's weight planes through LDS."""
    from chatterbox_vllm_amd.cfm_decoder import linear_op
    g = torch.Generator().manual_seed(Cin + Cout + K + L)
    x = torch.randn(2, Cin, L, generator=g)
    w = torch.randn(Cout, Cin, K, generator=g) / (Cin * K) ** 0.5
    b = torch.randn(Cout, generator=g)
    res = torch.randn(2, Cout, L, generator=g)
    gelu = Cout >= 1024
    want = torch.nn.functional.conv1d(torch.nn.functional.pad(x.double(), (pad, 0)), w.double(), b.double())
    if gelu:
        want = torch.nn.functional.gelu(want)
    want = want + res.double()
    ld = (L + 63) // 64 * 64 + 64
    xd = torch.zeros(2, Cin, ld, device="cuda")
    xd[:, :, L:] = 7.0                                        # columns past L must not be read as data
    xd[:, :, :L] = x.cuda()
    y, _, _ = linear_op(xd[:, :, :L], w, b, pad=pad, gelu=gelu, res=res.cuda(), split=split)
    assert (y.double().cpu() - want).abs().max().item() < 2e-5


@pytest.mark.parametrize("split", [False, True], ids=["fp32", "split_bf16"])
@pytest.mark.parametrize("Cout,L,offset", [(1536, 1000, 0.0), (1024, 333, 3.0), (1024, 4200, -1.0), (256, 40, 0.5), (1536, 2750, 0.7)])
def test_linear_kernel_with_the_layer_norm_folded_in(Cout, L, offset, split):
    """norm1 / norm3 of a transformer block ride in the layer that consumes them (LinArgs::rsum): W LayerNorm(x) + b against
    fp64, with channel means away from zero (the folded form subtracts mean * row-sum after the product), for every
    workgroup form, with GELU and with the frame-major part."""
    from chatterbox_vllm_amd.cfm_decoder import linear_op
    g = torch.Generator().manual_seed(Cout + L)
    x = 2.0 * torch.randn(2, 256, L, generator=g) + offset
    w = torch.randn(Cout, 256, 1, generator=g) / 16
    b = torch.randn(Cout, generator=g)
    gamma = 1.0 + 0.2 * torch.randn(256, generator=g)
    beta = 0.3 * torch.randn(256, generator=g)
    n = torch.nn.functional.layer_norm(x.double().transpose(1, 2), (256,), gamma.double(), beta.double(), 1e-5).transpose(1, 2)
    want = torch.nn.functional.conv1d(n, w.double(), b.double())
    gelu = Cout == 1024
    if gelu:
        want = torch.nn.functional.gelu(want)
    ld = (L + 63) // 64 * 64 + 64
    xd = torch.zeros(2, 256, ld, device="cuda")
    xd[:, :, :L] = x.cuda()
    y, vt, _ = linear_op(xd[:, :, :L], w, b, gelu=gelu, v_from=1024 if Cout == 1536 else None, ln=(gamma, beta), split=split)
    if Cout == 1536:
        assert (vt.double().cpu() - want[:, 1024:].transpose(1, 2)).abs().max().item() < 2e-5
        want = want[:, :1024]
    assert (y.double().cpu() - want).abs().max().item() < 2e-5


@pytest.mark.parametrize("split", [False, True], ids=["fp32", "split_bf16"])
def test_qkv_projection_writes_v_frame_major(split):
    from chatterbox_vllm_amd.cfm_decoder import linear_op
    L = 210
    g = torch.Generator().manual_seed(2)
    x = torch.randn(2, 256, L, generator=g)
    w = torch.randn(1536, 256, 1, generator=g) / 16
    want = torch.nn.functional.conv1d(x.double(), w.double())
    xd = torch.zeros(2, 256, 320, device="cuda")
    xd[:, :, :L] = x.cuda()
    y, vt, _ = linear_op(xd[:, :, :L], w, torch.zeros(1536), v_from=1024, split=split)
    assert y.shape == (2, 1024, L) and vt.shape == (2, L, 512)
    assert (y.double().cpu() - want[:, :1024]).abs().max().item() < 1e-5
    assert (vt.double().cpu() - want[:, 1024:].transpose(1, 2)).abs().max().item() < 1e-5


@pytest.mark.parametrize("split", [False, True])
@pytest.mark.parametrize("L", [1, 31, 32, 100, 129, 1000, 1100, 2000, 2101])
def test_attention_kernel_against_fp64(L, split):
    """softmax(q k^T / 8) v per head against fp64, lengths around the 32-key tile and the four-wave split, with score
    spreads of tens of units so that the running maximum moves; 1100, 2000 and 2101 frames make the fp32 launcher pick
    three, two and one query tiles per wave (1000: one tile on the fp32 form, two on the split form).  split: the form that holds every operand as three bf16 values and takes six partial
    products per product on the bf16 matrix instruction, to the same tolerance."""
    from chatterbox_vllm_amd.cfm_decoder import attention_op
    g = torch.Generator().manual_seed(L)
    q = 2.0 * torch.randn(2, 512, L, generator=g)
    k = 2.0 * torch.randn(2, 512, L, generator=g)
    v = torch.randn(2, 512, L, generator=g)
    o, _ = attention_op(q.cuda(), k.cuda(), v.cuda(), split=split)
    qh, kh, vh = (t.double().view(2, 8, 64, L) for t in (q, k, v))
    a = torch.softmax(qh.transpose(2, 3) @ kh / 8.0, dim=-1)                     # [2, 8, L, L]
    want = (vh @ a.transpose(2, 3)).reshape(2, 512, L)
    assert (o.double().cpu() - want).abs().max().item() < 2e-5


@pytest.mark.parametrize("split", [False, True])
def test_attention_kernel_with_several_utterances_and_their_own_key_counts(split):
    """Three utterances of 150, 97 and 33 real frames in one launch at L = 150: every query of an utterance sees only that
    utterance's keys (fp64 softmax over the first keys[u] frames); what lies past them in q, k, v is large on purpose, so a
    key that leaks would show.  Outputs of the padding queries only have to be finite."""
    from chatterbox_vllm_amd.cfm_decoder import attention_op
    L, keys = 150, [150, 97, 33]
    g = torch.Generator().manual_seed(77)
    q = 2.0 * torch.randn(6, 512, L, generator=g)
    k = 2.0 * torch.randn(6, 512, L, generator=g)
    v = torch.randn(6, 512, L, generator=g)
    for u, n in enumerate(keys):
        k[2 * u:2 * u + 2, :, n:] *= 5.0
        v[2 * u:2 * u + 2, :, n:] += 100.0
    o, _ = attention_op(q.cuda(), k.cuda(), v.cuda(), keys=keys, split=split)
    assert torch.isfinite(o).all()
    for u, n in enumerate(keys):
        r = slice(2 * u, 2 * u + 2)
        qh, kh, vh = (t[r, :, :n].double().reshape(2, 8, 64, n) for t in (q, k, v))
        a = torch.softmax(qh.transpose(2, 3) @ kh / 8.0, dim=-1)
        want = (vh @ a.transpose(2, 3)).reshape(2, 512, n)
        assert (o[r, :, :n].double().cpu() - want).abs().max().item() < 2e-5


@pytest.fixture(scope="module", **_FORMS)
def cfm4(sd, request):
    from chatterbox_vllm_amd.cfm_decoder import CausalConditionalCFM
    m = CausalConditionalCFM(sd, max_frames=300, generator=torch.Generator().manual_seed(3), max_batch=4, split_bf16=request.param)
    yield m
    m.shutdown()


def test_batched_solve_gives_every_utterance_its_own_result(cfm4, sd):
    """Four utterances of different lengths through shared launches: each result is held to the fp64 restatement of that
    utterance alone (the bar of the single-utterance solve) and sits within fp32 rounding of the single-utterance call."""
    from oracle import cfm_oracle as O
    frames, prompts = [96, 131, 40, 70], [30, 50, 8, 20]
    ins = [CW.synthetic_inputs(n, p, seed=20 + i) for i, (n, p) in enumerate(zip(frames, prompts))]
    L = max(frames)
    pad = lambda t: torch.nn.functional.pad(t, (0, L - t.shape[2]))
    ts = CW.t_span(10)
    got = cfm4.solve_euler_batch(torch.cat([pad(d["noise"]) for d in ins]), ts, torch.cat([pad(d["mu"]) for d in ins]),
                                 torch.cat([d["spks"] for d in ins]), torch.cat([pad(d["cond"]) for d in ins]), frames)
    assert got.shape == (4, 80, L)
    for u, d in enumerate(ins):
        mine = got[u:u + 1, :, :frames[u]]
        want = O.solve_euler(sd, d["noise"], ts, d["mu"], d["spks"], d["cond"], dtype=torch.float64)
        assert _rel(mine, want) < 1e-4
        single = cfm4.solve_euler(d["noise"], ts, d["mu"], d["spks"], d["cond"])
        assert _rel(mine, single.double().cpu()) < 2e-5
    # a batch of one is the single call, bit for bit; a second batch after a longer one leaves no trace
    one = cfm4.solve_euler_batch(ins[2]["noise"], ts, ins[2]["mu"], ins[2]["spks"], ins[2]["cond"], [frames[2]])
    assert torch.equal(one, cfm4.solve_euler(ins[2]["noise"], ts, ins[2]["mu"], ins[2]["spks"], ins[2]["cond"]))
    again = cfm4.solve_euler_batch(torch.cat([pad(d["noise"]) for d in ins]), ts, torch.cat([pad(d["mu"]) for d in ins]),
                                   torch.cat([d["spks"] for d in ins]), torch.cat([pad(d["cond"]) for d in ins]), frames)
    for u in range(4):
        assert torch.equal(again[u, :, :frames[u]], got[u, :, :frames[u]])


def test_batched_solve_limits(cfm4):
    d = CW.synthetic_inputs(50, 10)
    five = lambda t: torch.cat([t] * 5)
    with pytest.raises(ValueError):
        cfm4.solve_euler_batch(five(d["noise"]), CW.t_span(2), five(d["mu"]), five(d["spks"]), five(d["cond"]), [50] * 5)


def test_chain_on_several_utterances_at_once(sd):
    """S3Token2Wav.inference_batch: three utterances of one voice through the flow-matching stage together give the
    waveforms of three inference calls, to the rounding the shared launches allow on the mel (2e-5 of its range) and what
    the vocoder makes of that."""
    from chatterbox_vllm_amd import flow_weights as FW, hift_weights as HW
    from chatterbox_vllm_amd.s3gen import S3Token2Wav
    flow_sd = dict(FW.synthetic_state_dict())
    flow_sd.update(sd)
    torch.manual_seed(11)
    s3 = S3Token2Wav(flow_sd, HW.synthetic_state_dict(), max_tokens=64, max_batch=3)
    g = torch.Generator().manual_seed(8)
    ref = {"prompt_token": FW.synthetic_tokens(14, seed=3).view(1, -1),
           "prompt_feat": 2.0 * torch.randn(1, 28, 80, generator=g) - 4.0,
           "embedding": torch.randn(1, 192, generator=g)}
    toks = [FW.synthetic_tokens(n, seed=40 + n) for n in (30, 12, 45)]
    ins = [s3.encoder.decoder_inputs(t.view(1, -1), ref["prompt_token"], ref["prompt_feat"], ref["embedding"]) for t in toks]
    mels = s3.decoder.forward_batch([d["mu"] for d in ins], [d["spks"] for d in ins], [d["cond"] for d in ins])
    for d, t, mel in zip(ins, toks, mels):
        one, _ = s3.decoder(mu=d["mu"], mask=d["mask"], spks=d["spks"], cond=d["cond"], n_timesteps=10)
        assert mel.shape == one.shape == (1, 80, 28 + 2 * t.numel())
        assert _rel(mel, one.double().cpu()) < 2e-5
    wavs = s3.inference_batch(toks, ref, seed=3)
    for t, wav in zip(toks, wavs):
        assert wav.shape == (1, 2 * t.numel() * 480) and torch.isfinite(wav).all() and not wav[0, :480].any()
    s3.shutdown()


def test_two_chains_in_flight_give_the_bits_of_one_at_a_time(sd):
    """Several utterances in flight (bench.py's s3gen_job leg): two token -> waveform chains, each with its own handle set,
    stream and host thread, running at the same time return bit for bit what one chain returns for the same utterances one
    after the other."""
    import threading
    from chatterbox_vllm_amd import flow_weights as FW, hift_weights as HW
    from chatterbox_vllm_amd.s3gen import S3Token2Wav
    flow_sd = dict(FW.synthetic_state_dict())
    flow_sd.update(sd)
    hift_sd = HW.synthetic_state_dict()
    g = torch.Generator().manual_seed(21)
    ref = {"prompt_token": FW.synthetic_tokens(20, seed=3).view(1, -1), "prompt_feat": 2.0 * torch.randn(1, 40, 80, generator=g) - 4.0,
           "embedding": torch.randn(1, 192, generator=g)}
    toks = [FW.synthetic_tokens(n, seed=30 + n) for n in (35, 61, 18, 50)]
    chains = []
    for _ in range(2):
        torch.manual_seed(9)                                     # both chains draw the same fixed noise
        chains.append(S3Token2Wav(flow_sd, hift_sd, max_tokens=96))
    want = [chains[0].inference(t, ref, n_timesteps=3, seed=5)[0].cpu() for t in toks]
    got = [None] * len(toks)
    streams = [torch.cuda.Stream() for _ in range(2)]

    def work(w):
        with torch.cuda.stream(streams[w]):
            for i in range(w, len(toks), 2):
                got[i] = chains[w].inference(toks[i], ref, n_timesteps=3, seed=5)[0]
            streams[w].synchronize()

    threads = [threading.Thread(target=work, args=(w,)) for w in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for a, b in zip(got, want):
        assert torch.equal(a.cpu(), b)
    for c in chains:
        c.shutdown()
