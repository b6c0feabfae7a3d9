"""Host-side logic on the CPU: text normalisation, tokenizers, id post-processing, the
conditioning encoder, checkpoint routing, and request sharding over a world-size-2 gloo group.
Expected values come from tests/golden (captured from the reference by make_reference_fixtures.py)."""
import json
import os

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from chatterbox_vllm_amd import cond_enc as CE
from chatterbox_vllm_amd import weights as W
from chatterbox_vllm_amd.config import T3ModelConfig
from chatterbox_vllm_amd.text_utils import punc_norm, guard_cap, batch_max_tokens, nonspace_chars

REF_EN_VOCAB = "/root/reference/src/chatterbox_vllm/models/t3/tokenizer.json"
REF_MTL_VOCAB = "/root/reference/t3-multilingual-model/grapheme_mtl_merged_expanded_v1.json"


@pytest.fixture(scope="module")
def text_fx(golden_dir):
    with open(os.path.join(golden_dir, "ref_text_fixture.json")) as f:
        return json.load(f)


def test_punc_norm_matches_reference(text_fx):
    for src, want in text_fx["punc_norm"]:
        assert punc_norm(src) == want


def test_benchmark_job_histogram(text_fx):
    b = text_fx["benchmark_text_1"]
    lens = b["text_id_counts"]
    assert b["n_chunks"] == len(lens) == 154
    assert min(lens) == 17 and max(lens) == 484 and abs(sum(lens) / len(lens) - 151.7) < 0.1


def test_max_tokens_rule():
    # ceil(clamp(ceil(chars*2.2), 64, 1200) * 1.6) per prompt; batch cap = min over prompts + 16, capped
    p_short = "[START]Hi.[STOP]"
    p_long = "[START]" + "a" * 300 + ".[STOP]"
    assert nonspace_chars(p_short) == 3
    assert guard_cap(p_short) == 103            # max(64, ceil(6.6)) = 64 -> ceil(102.4)
    assert guard_cap(p_long) == int(np.ceil(int(np.ceil(301 * 2.2)) * 1.6))
    assert batch_max_tokens([p_short, p_long], 1000, 1000) == 103 + 16
    assert batch_max_tokens([p_long], 1000, 400) == 400
    assert batch_max_tokens([p_long], 50, 1000) == 50


def test_postprocess_speech_ids():
    from chatterbox_vllm_amd.tts import postprocess_speech_ids, STOP_OFFSET_ID
    S = STOP_OFFSET_ID
    assert postprocess_speech_ids([1005, 1006, 1007, 1008, S, 1009]).tolist() == [5, 6]
    assert postprocess_speech_ids([1005, 7561, 1006, 9000]).tolist() == [6]
    assert postprocess_speech_ids([1001, S]).tolist() == []
    assert postprocess_speech_ids([S]).dtype == torch.int64
    many = [1000 + (i % 6000) for i in range(500)]
    assert len(postprocess_speech_ids(many, prompt="[START]Hi.[STOP]")) == 103          # per-prompt guard


@pytest.mark.skipif(not os.path.isfile(REF_EN_VOCAB), reason="reference vocabulary file not present on this machine")
def test_tokenizers_match_reference(text_fx):
    from chatterbox_vllm_amd.tokenizer import EnTokenizer, MTLTokenizer
    tok = EnTokenizer(REF_EN_VOCAB)
    for s, ids in text_fx["en"]:
        assert tok(s) == ids
    assert tok.vocab_size == text_fx["vocab_sizes"]["en"] == 704
    mtl = MTLTokenizer(REF_MTL_VOCAB)
    for s, ids in text_fx["mtl_fr"]:
        assert mtl("[fr]" + s) == ids                     # leading tag form (mtl_tts.py:423)
        assert mtl(s, language_id="fr") == ids            # tokenization_kwargs form
        assert ids[0] == text_fx["mtl_special_ids"]["[fr]"] and ids[1] == text_fx["mtl_special_ids"]["[START]"]
    assert mtl("[xx]abc") != mtl("abc")                   # unknown tags stay text


def test_korean_jamo_decomposition():
    from chatterbox_vllm_amd.tokenizer import korean_to_jamo
    assert korean_to_jamo("한") == "한"
    assert korean_to_jamo("가a") == "가a"


def test_tokenizer_lookup_fails_loudly(tmp_path, monkeypatch):
    from chatterbox_vllm_amd.tokenizer import EnTokenizer
    monkeypatch.chdir(tmp_path)
    with pytest.raises(FileNotFoundError):
        EnTokenizer.from_pretrained(str(tmp_path))


def test_cond_encoder_oracle_matches_reference(golden_dir):
    from oracle import cond_enc_oracle as CEO
    d = np.load(os.path.join(golden_dir, "ref_plugin_fixture.npz"))
    w = {k[len("cond_enc."):]: v for k, v in CE.synthetic_weights(1234).items()}
    g = torch.Generator().manual_seed(31)
    spk = torch.randn(1, 256, generator=g)
    prompt = torch.randn(150, 1024, generator=g) * 0.05
    for e in (0.5, 0.8):
        got = CEO.encode(w, spk, prompt, e)
        assert got.shape == (34, 1024)
        assert torch.allclose(got, torch.from_numpy(d[f"condenc_out_{e}"]), atol=1e-5, rtol=1e-5)
    assert torch.allclose(CEO.emotion_row(w, 0.3), torch.from_numpy(d["condenc_emotion_row_0.3"]), atol=1e-7)


def test_checkpoint_routing():
    cfg = T3ModelConfig.english(n_layers=1)
    sd = W.synthetic_state_dict(cfg)
    assert sd["speech_head.weight"].shape == (8194, 1024) and sd["speech_head.weight"].dtype == torch.bfloat16
    extra = dict(sd)
    extra["text_head.weight"] = torch.zeros(704, 1024)          # skipped like t3.py:363-366
    extra["cond_enc.spkr_enc.bias"] = torch.zeros(1024)
    keep, skipped = W.route_checkpoint(extra.items(), cfg)
    assert set(keep) == set(sd) and sorted(skipped) == ["cond_enc.spkr_enc.bias", "text_head.weight"]
    with pytest.raises(KeyError):
        W.route_checkpoint([(k, v) for k, v in sd.items() if k != "tfmr.norm.weight"], cfg)
    bad = dict(sd)
    bad["text_emb.weight"] = torch.zeros(700, 1024)
    with pytest.raises(ValueError):
        W.route_checkpoint(bad.items(), cfg)
    full = T3ModelConfig.english()
    assert full.backbone_params == 503378944 and full.weight_bytes_per_step == 1023539200
    assert full.kv_bytes_per_token == 122880
    # the tables come first in the generator walk, so they do not depend on the layer count
    assert torch.equal(W.synthetic_state_dict(T3ModelConfig.english(n_layers=0))["speech_emb.weight"], sd["speech_emb.weight"])


# ---- request sharding, world size 2 over gloo ------------------------------------------------
class _FakeEngine:
    """Deterministic stand-in: ids depend only on the request, like the real engine's greedy path."""

    def generate(self, items, sps):
        from chatterbox_vllm_amd.engine import RequestOutput, CompletionOutput
        outs = []
        for it, sp in zip(items, sps):
            ids = [1000 + (7 * t + k) % 6000 for k, t in enumerate(it["prompt_token_ids"][: sp.max_tokens])]
            outs.append(RequestOutput("x", None, it["prompt_token_ids"], [CompletionOutput(0, ids, "length")]))
        return outs


def _worker(rank, world, port, q):
    import torch.distributed as dist
    from chatterbox_vllm_amd.engine import SamplingParams
    from chatterbox_vllm_amd import sharding as S
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cfg = T3ModelConfig.english(n_layers=0)
        sd = W.synthetic_state_dict(cfg, seed=1234 if rank == 0 else 99)        # only rank 0 holds the real values
        got = S.broadcast_state_dict(sd, src=0)
        want = W.synthetic_state_dict(cfg, seed=1234)
        ok_bcast = all(torch.equal(got[k], want[k]) for k in want)
        items = [{"prompt_token_ids": list(range(3 + i, 3 + i + 4 + (i % 5)))} for i in range(7)]
        sps = [SamplingParams(max_tokens=3 + (i % 4)) for i in range(7)]
        res = S.generate_sharded(_FakeEngine(), items, sps)
        ref = [o.outputs[0].token_ids for o in _FakeEngine().generate(items, sps)]
        ragged = S.gather_token_ids([[rank] * (rank + 1), []] if rank == 0 else [[5, 6, 7]])
        q.put((rank, ok_bcast, res == ref, ragged, S.shard_indices(7, world, rank)))
    finally:
        dist.destroy_process_group()


def test_sharding_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p_ in procs:
        p_.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p_ in procs:
        p_.join(60)
        assert p_.exitcode == 0
    for rank, ok_bcast, ok_gen, ragged, idx in res:
        assert ok_bcast and ok_gen
        assert ragged == [[0], [], [5, 6, 7]]
    assert res[0][4] == [0, 2, 4, 6] and res[1][4] == [1, 3, 5]


def test_benchmark_job_builder_follows_the_reference_histogram(text_fx):
    """bench.py's synthetic job: 154 requests with the text-id counts of the reference's benchmark input, output
    lengths proportional to them with the README-implied mean, everything inside max_model_len; deterministic."""
    import bench
    job = bench.build_job(seed=7)
    lens = text_fx["benchmark_text_1"]["text_id_counts"]
    assert [len(t) for t, _, _ in job] == [max(3, n) for n in lens] and len(job) == 154
    assert all(t[0] == 255 and t[-1] == 0 and all(0 <= i < 704 for i in t) for t, _, _ in job)
    assert all(16 <= o and 34 + len(t) + 1 + o <= bench.MAX_MODEL_LEN for t, o, _ in job)
    mean_out = sum(o for _, o, _ in job) / len(job)
    assert abs(mean_out - bench.MEAN_OUT) < 25
    assert bench.build_job(seed=7) == job and bench.build_job(seed=8) != job
    single = bench.build_job(seed=7, workload="single")
    assert len(single) == 1 and 34 + len(single[0][0]) + 1 + single[0][1] == bench.MAX_MODEL_LEN
    mtl = bench.build_job(seed=7, workload="multilingual")
    assert len(mtl) == 128 and {v for _, _, v in mtl} == {0, 1, 2, 3} and {t[0] for t, _, _ in mtl} == {634, 725}


def test_bench_starts_its_own_ranks_from_a_plain_shell():
    """`python bench.py --gpus 2` without torchrun around it: the script starts the ranks itself (one child
    torch.distributed.run, before any GPU call), rank 0's JSON line comes back on stdout, and `ranks_seen` (an
    all-reduce over the group) says both ranks took part.  --rehearse swaps the engine for a stand-in so this runs
    on a machine without a GPU; weak scaling (every rank its own job) and strong scaling (one job dealt out)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    for extra, per_rank in (([], [6, 6]), (["--scaling", "strong"], [4, 3])):
        p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--rehearse", "--steps", "1",
                            "--warmup", "0", "--requests", "7" if extra else "6"] + extra,
                           capture_output=True, text=True, timeout=300, env=env, cwd=root)
        assert p.returncode == 0, p.stderr[-2000:]
        lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1
        rec = json.loads(lines[0])
        assert rec["n_gpus"] == 2 and rec["ranks_seen"] == 2 and rec["scaling"] == ("strong" if extra else "weak")
        assert rec["config"]["requests_per_gpu"] == per_rank[0] and "rehearsal" in rec["data"]
        assert rec["value"] > 0 and rec["config"]["tokens_per_step_per_gpu"] == 4 * per_rank[0]
    # a rank count that disagrees with the launcher's world is an error, not an assertion trace
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--rehearse"],
                       capture_output=True, text=True, timeout=120, cwd=root,
                       env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"))
    assert p.returncode == 0 and json.loads(p.stdout.strip().splitlines()[-1])["ranks_seen"] == 1


def _write_checkpoint_dir(d, golden_dir, n_layers=1):
    """A checkpoint directory of the shape ChatterboxTTS.from_local reads (reference: tts.py:91-158): the T3
    safetensors file (seeded synthetic values), tokenizer.json (a tiny BPE vocabulary made for these tests) and
    conds.pt (tensors only)."""
    import shutil
    from safetensors.torch import save_file
    cfg = T3ModelConfig.english(n_layers=n_layers)
    sd = W.synthetic_state_dict(cfg)
    sd.update({k: v for k, v in CE.synthetic_weights(1234).items()})
    sd["text_head.weight"] = torch.zeros(704, 1024, dtype=torch.bfloat16)        # present in real checkpoints, skipped
    save_file({k: v.contiguous() for k, v in sd.items()}, os.path.join(d, "t3_cfg.safetensors"))
    shutil.copy(os.path.join(golden_dir, "tiny_tokenizer.json"), os.path.join(d, "tokenizer.json"))
    torch.save({"t3": {"speaker_emb": torch.zeros(1, 256), "cond_prompt_speech_tokens": torch.zeros(1, 150, dtype=torch.long),
                       "emotion_adv": 0.5 * torch.ones(1, 1, 1)}, "gen": {}}, os.path.join(d, "conds.pt"))
    return cfg, sd


def test_from_local_reads_a_checkpoint_directory(tmp_path, golden_dir, monkeypatch):
    """ChatterboxTTS.from_local on a directory: safetensors load, tokenizer lookup in that directory, weights-only
    load of conds.pt, the layer count taken from the checkpoint, kwargs passed through to the engine.  The engine
    class is replaced by a recorder here (no GPU on this machine); tests/test_facade_gpu.py runs the same directory
    through the real engine."""
    from chatterbox_vllm_amd import tts as T
    cfg, sd = _write_checkpoint_dir(str(tmp_path), golden_dir, n_layers=1)
    seen = {}

    class Recorder:
        def __init__(self, model_config, state_dict, tokenizer=None, **kw):
            seen.update(cfg=model_config, sd=state_dict, tok=tokenizer, kw=kw)
            self.device = 0
    monkeypatch.setattr(T, "T3Engine", Recorder)
    m = T.ChatterboxTTS.from_local(str(tmp_path), max_model_len=300, max_batch_size=3, compile=True,
                                   gpu_memory_utilization=0.4, scheduling_policy="fcfs")
    assert seen["cfg"].n_layers == 1 and seen["cfg"].text_vocab == 704
    assert torch.equal(seen["sd"]["speech_head.weight"], sd["speech_head.weight"]) and "text_head.weight" in seen["sd"]
    assert seen["kw"]["max_model_len"] == 300 and seen["kw"]["max_num_seqs"] == 3 and seen["kw"]["scheduling_policy"] == "fcfs"
    assert "gpu_memory_utilization" not in seen["kw"]
    assert seen["tok"]("[START]Hi there.[STOP]") == [255, 36, 11, 2, 62, 20, 7, 55, 0]
    assert m.default_conds is not None and m.default_conds["t3"]["speaker_emb"].shape == (1, 256)
    assert m.cond_enc_weights is not None and "spkr_enc.weight" in m.cond_enc_weights
    assert m.sr == 24000 and m.max_model_len == 300
    with pytest.raises(FileNotFoundError):
        T.ChatterboxTTS.from_local(str(tmp_path / "nowhere"))
    os.remove(os.path.join(str(tmp_path), "tokenizer.json"))
    monkeypatch.chdir(tmp_path)
    with pytest.raises(FileNotFoundError):
        T.ChatterboxTTS.from_local(str(tmp_path))


def test_waveform_post_processing_follows_the_reference(monkeypatch):
    """tts.py:495-541: the waveform is cut to 960 samples per speech token, then after the last 50 ms window (20 ms hop) whose
    RMS is within 35 dB of the loudest window, plus 50 ms."""
    from chatterbox_vllm_amd.tts import postprocess_waveform
    sr = 24000
    g = torch.Generator().manual_seed(0)
    wav = torch.zeros(1, 30 * 960 + 500)
    wav[0, : 12000] = 0.3 * torch.randn(12000, generator=g)                 # half a second of signal, then silence
    out = postprocess_waveform(wav, 30, sr)
    # the last active window starts at hop index 24 (24 * 480 = 11520 < 12000): cut = 25 * 480 + 1200
    assert out.shape == (1, 25 * 480 + 1200) and torch.equal(out, wav[:, : out.shape[1]])
    loud = 0.3 * torch.randn(1, 30 * 960 + 500, generator=g)
    assert postprocess_waveform(loud, 30, sr).shape == (1, 30 * 960)      # alignment cap only
    monkeypatch.setenv("CHATTERBOX_ALIGN_HARD", "0")
    monkeypatch.setenv("CHATTERBOX_TAIL_TRIM", "0")
    assert postprocess_waveform(wav, 30, sr).shape == wav.shape
    assert postprocess_waveform(torch.zeros(1, 0), 0, sr).numel() == 0
