#!/usr/bin/env python3
"""Capture golden vectors from the reference's own T3 plugin functions.

Run in the build container only (needs /root/reference):

    python tests/golden/make_reference_fixtures.py

It imports ``src/chatterbox_vllm/models/t3/t3.py`` from the reference tree with the
absent third-party ``vllm`` package replaced by empty stand-in modules (the plugin
only subclasses / decorates with those names at import time), builds a
``T3VllmModel`` without running its vLLM-dependent ``__init__`` and calls the
reference's ``create_triangular_matrix``, ``get_input_embeddings`` (whole prompt,
chunked prompt, prompt followed by decode ids) and ``compute_logits``.  Inputs come
from the seeded synthetic checkpoint, so the test can regenerate them; only the
reference's OUTPUTS are stored, in ``ref_plugin_fixture.npz``.  It also records the
reference tokenizer's ids for the example prompts and the text-length histogram of
the benchmark input (``ref_text_fixture.json``).
"""
import importlib
import json
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..", "..")))

from chatterbox_vllm_amd.config import T3ModelConfig  # noqa: E402
from chatterbox_vllm_amd import weights as W  # noqa: E402


def _stub_vllm():
    names = {
        "vllm": [],
        "vllm.config": ["VllmConfig", "ModelConfig"],
        "vllm.model_executor": [],
        "vllm.model_executor.layers": [],
        "vllm.model_executor.layers.logits_processor": ["LogitsProcessor"],
        "vllm.model_executor.layers.vocab_parallel_embedding": ["ParallelLMHead"],
        "vllm.model_executor.models": [],
        "vllm.model_executor.models.interfaces": ["MultiModalEmbeddings", "SupportsMultiModal"],
        "vllm.model_executor.models.interfaces_base": ["VllmModelForTextGeneration"],
        "vllm.model_executor.models.llama": ["LlamaModel"],
        "vllm.model_executor.sampling_metadata": ["SamplingMetadata"],
        "vllm.multimodal": ["MULTIMODAL_REGISTRY"],
        "vllm.multimodal.inputs": ["MultiModalKwargs", "MultiModalKwargsItem", "MultiModalBatchedField"],
        "vllm.multimodal.parse": ["MultiModalDataParser", "ModalityDataItems"],
        "vllm.multimodal.processing": ["BaseMultiModalProcessor", "BaseProcessingInfo", "MultiModalDataDict",
                                       "MultiModalDataItems", "MultiModalFieldConfig", "PromptUpdate",
                                       "MultiModalInputs", "PlaceholderRange"],
        "vllm.multimodal.profiling": ["BaseDummyInputsBuilder"],
        "vllm.sequence": ["IntermediateTensors"],
    }

    class _Generic:
        def __class_getitem__(cls, item):
            return cls

    class _Registry:
        def register_processor(self, *a, **k):
            return lambda c: c

    for mod, attrs in names.items():
        m = types.ModuleType(mod)
        m.__path__ = []
        for a in attrs:
            setattr(m, a, _Registry() if a == "MULTIMODAL_REGISTRY" else type(a, (_Generic,), {}))
        sys.modules[mod] = m


def _import_reference_t3():
    _stub_vllm()
    src = os.path.join(REF, "src")
    for name, rel in [("chatterbox_vllm", "chatterbox_vllm"),
                      ("chatterbox_vllm.models", "chatterbox_vllm/models"),
                      ("chatterbox_vllm.models.t3", "chatterbox_vllm/models/t3"),
                      ("chatterbox_vllm.models.t3.modules", "chatterbox_vllm/models/t3/modules")]:
        pkg = types.ModuleType(name)
        pkg.__path__ = [os.path.join(src, rel)]   # package shell: sub-modules import, __init__ is not run
        sys.modules[name] = pkg
    return importlib.import_module("chatterbox_vllm.models.t3.t3")


def _make_model(t3mod, cfg, cfg_scale):
    from chatterbox_vllm.models.t3.modules.t3_config import T3Config
    sd = W.synthetic_state_dict(T3ModelConfig(n_layers=0, text_vocab=cfg.text_vocab))
    m = object.__new__(t3mod.T3VllmModel)
    torch.nn.Module.__init__(m)
    m.t3conf = T3Config.multilingual() if cfg.is_multilingual else T3Config()
    m.dim = 1024
    m.text_emb = torch.nn.Embedding(cfg.text_vocab, 1024, dtype=torch.bfloat16)
    m.speech_emb = torch.nn.Embedding(cfg.speech_vocab, 1024, dtype=torch.bfloat16)
    m.text_emb.weight.data.copy_(sd["text_emb.weight"])
    m.speech_emb.weight.data.copy_(sd["speech_emb.weight"])
    m.precomputed_text_pos_emb = sd["text_pos_emb.emb.weight"].clone()
    m.precomputed_speech_pos_emb = sd["speech_pos_emb.emb.weight"].clone()
    m.speech_head = torch.nn.Linear(1024, cfg.speech_vocab, bias=False, dtype=torch.bfloat16)
    m.speech_head.weight.data.copy_(sd["speech_head.weight"])
    # stand-in for vLLM's LogitsProcessor: fp32-accumulated matmul rounded to the model dtype
    m.logits_processor = lambda head, h, _md: (h.float() @ head.weight.float().T).to(torch.bfloat16)
    m.cfg_scale = cfg_scale
    return m


def _bits(t):
    """bf16 bit patterns (uint16) of a tensor whose values are exactly representable in bf16."""
    f = t.float()
    b = f.to(torch.bfloat16)
    assert torch.equal(b.float(), f) or bool(torch.isinf(f).any()), "value not representable in bf16"
    return b.view(torch.int16).numpy().view(np.uint16)


def main():
    t3mod = _import_reference_t3()
    out = {}
    with torch.inference_mode():
        for tag, cfg, text_ids in [
            ("en", T3ModelConfig.english(), [255, 33, 14, 2, 71, 200, 9, 0]),
            ("mtl", T3ModelConfig.multilingual(), [634, 255, 1200, 2, 2453, 17, 1001, 0]),
        ]:
            m = _make_model(t3mod, cfg, cfg_scale=0.5 if tag == "en" else 3.0)
            cond = W.synthetic_cond_emb(voice=3)
            T = len(text_ids)
            prompt = [t3mod.PREFILL_COND_START_TOKEN, *([text_ids[0]] * 32), t3mod.PREFILL_COND_END_TOKEN,
                      *text_ids, t3mod.PREFILL_END_TOKEN]
            tri = t3mod.create_triangular_matrix(T, 1024)
            mm = torch.cat([cond, tri, torch.zeros(1, 1024)], 0)
            ids = torch.tensor(prompt)
            full = m.get_input_embeddings(ids, [mm])
            out[f"{tag}_text_ids"] = np.array(text_ids, dtype=np.int32)
            out[f"{tag}_prompt_ids"] = np.array(prompt, dtype=np.int32)
            out[f"{tag}_tri_rowsum"] = tri.sum(1).numpy()
            out[f"{tag}_embeds_full_dtype"] = str(full.dtype)   # fp32: cat(fp32 cond rows, bf16 text rows) promotes
            out[f"{tag}_embeds_full"] = full.float().numpy()
            # prompt split at several cuts reproduces the whole-prompt rows (recorded as a fact)
            cuts, same = (1, 10, 33, 34, 35, 38, len(prompt) - 1), []
            for cut in cuts:
                a = m.get_input_embeddings(ids[:cut], [mm[:cut]])
                b = m.get_input_embeddings(ids[cut:], [mm[cut:]])
                same.append(bool(torch.equal(torch.cat([a, b], 0).float(), full.float())))
            out[f"{tag}_cuts"] = np.array(cuts, dtype=np.int32)
            out[f"{tag}_cut_equals_full"] = np.array(same)
            # a whole prompt followed by decode ids of other requests (the supported order)
            dec = torch.tensor([1000 + 17, 1000 + 6561, 1000 + 8193])
            mixed = m.get_input_embeddings(torch.cat([ids, dec]), [mm])
            out[f"{tag}_decode_ids"] = dec.numpy().astype(np.int32)
            out[f"{tag}_mixed_head_equals_full"] = np.array(bool(torch.equal(mixed[:len(prompt)].float(), full.float())))
            out[f"{tag}_embeds_mixed_tail"] = _bits(mixed[len(prompt):])
            out[f"{tag}_embeds_decode_only"] = _bits(m.get_input_embeddings(dec, None))
            # CFG logits from given hidden states
            g = torch.Generator().manual_seed(77)
            hid = torch.randn(3, 2048, generator=g).to(torch.bfloat16)
            lg = m.compute_logits(hid, None)
            out[f"{tag}_hidden"] = _bits(hid)
            out[f"{tag}_logits"] = _bits(lg)
            out[f"{tag}_cfg_scale"] = np.float32(m.cfg_scale)
    # ---- conditioning encoder: reference T3CondEnc on the seeded cond_enc.* weights -------
    from chatterbox_vllm.models.t3.modules.cond_enc import T3Cond, T3CondEnc
    from chatterbox_vllm.models.t3.modules.t3_config import T3Config
    from chatterbox_vllm_amd import cond_enc as CE
    with torch.inference_mode():
        enc = T3CondEnc(T3Config()).eval()
        enc.load_state_dict({k[len("cond_enc."):]: v for k, v in CE.synthetic_weights(1234).items()})
        g = torch.Generator().manual_seed(31)
        spk = torch.randn(1, 256, generator=g)
        prompt = torch.randn(150, 1024, generator=g) * 0.05
        for e in (0.5, 0.8):
            ref = enc(T3Cond(speaker_emb=spk, cond_prompt_speech_tokens=torch.zeros(1, 150, dtype=torch.long),
                             cond_prompt_speech_emb=prompt, emotion_adv=e * torch.ones(1, 1)))
            out[f"condenc_out_{e}"] = ref.float().numpy().astype(np.float32)
        out["condenc_emotion_row_0.3"] = enc.emotion_adv_fc(0.3 * torch.ones(1, 1))[0].float().numpy()
    np.savez_compressed(os.path.join(HERE, "ref_plugin_fixture.npz"), **out)
    print("wrote ref_plugin_fixture.npz:", {k: v.shape for k, v in out.items() if hasattr(v, "shape")})

    # ---- text side: tokenizer ids and benchmark length histogram ---------------
    from tokenizers import Tokenizer
    sys.path.insert(0, os.path.join(REF, "src", "chatterbox_vllm"))
    text_utils = importlib.import_module("text_utils")
    en_tok = Tokenizer.from_file(os.path.join(REF, "src/chatterbox_vllm/models/t3/tokenizer.json"))
    mtl_tok = Tokenizer.from_file(os.path.join(REF, "t3-multilingual-model/grapheme_mtl_merged_expanded_v1.json"))
    example_prompts = [
        "You are listening to a demo of the Chatterbox TTS model running on VLLM.",
        "This is a separate prompt to test the batching implementation.",
        "And here is a third prompt. It's a bit longer than the first one, but not by much.",
    ]
    tj = {"punc_norm": [], "en": [], "mtl_fr": []}
    for p in example_prompts + ["", "hello world", "Wait... what — really?; yes: “ok”"]:
        tj["punc_norm"].append([p, text_utils.punc_norm(p)])
    for p in example_prompts:
        s = "[START]" + text_utils.punc_norm(p) + "[STOP]"
        tj["en"].append([s, en_tok.encode(s.replace(" ", "[SPACE]")).ids])
    from unicodedata import normalize
    for p in ["Bonjour, comment ça va? Ceci est le modèle de synthèse vocale multilingue Chatterbox."]:
        s = "[START]" + text_utils.punc_norm(p) + "[STOP]"
        pre = normalize("NFKD", s.lower()).replace("[start]", "[START]").replace("[stop]", "[STOP]")
        tj["mtl_fr"].append([s, mtl_tok.encode(("[fr]" + pre).replace(" ", "[SPACE]")).ids])
    tj["special_ids"] = {k: en_tok.token_to_id(k) for k in ["[START]", "[STOP]", "[UNK]", "[SPACE]"]}
    tj["mtl_special_ids"] = {k: mtl_tok.token_to_id(k) for k in ["[START]", "[STOP]", "[fr]", "[en]", "[zh]"]}
    tj["vocab_sizes"] = {"en": en_tok.get_vocab_size(), "mtl": mtl_tok.get_vocab_size()}

    # benchmark.py:26-53,56-70 chunking of docs/benchmark-text-1.txt -> per-chunk id counts
    import re

    def split_text_by_sentence(text, max_chunk):
        sentences = text.split(". ")
        n_chunks = len(text) // max_chunk + 1
        approx = len(text) // n_chunks
        chunks, cur, cur_len = [], [], 0
        for s in sentences:
            s = " ".join(s.split()).strip()
            if cur_len + len(s) > approx:
                chunks.append(". ".join(cur))
                cur, cur_len = [s], len(s)
            else:
                cur.append(s)
                cur_len += len(s)
        if cur:
            chunks.append(". ".join(cur))
        return [c + "." if re.match(r"[a-zA-Z0-9]", c[-1]) else c for c in chunks if len(c) > 0]

    with open(os.path.join(REF, "docs/benchmark-text-1.txt")) as f:
        text = f.read()
    lines = [l.strip() for l in text.split("\n") if not l.startswith("#") and l.strip()]
    chunks = [c for l in lines for c in split_text_by_sentence(l, 400)]
    lens, chars = [], []
    for c in chunks:
        s = "[START]" + text_utils.punc_norm(c) + "[STOP]"
        lens.append(len(en_tok.encode(s.replace(" ", "[SPACE]")).ids))
        chars.append(sum(1 for ch in re.sub(r"\[[^\]]+\]", "", s) if not ch.isspace()))
    tj["benchmark_text_1"] = {"n_chunks": len(chunks), "text_id_counts": lens, "nonspace_chars": chars}

    # benchmark-multilingual-fr.py / -zh.py: same chunker (400 characters at ". " / 50 characters at the
    # ideographic full stop), ids from the multilingual vocabulary as mtl_tokenizer.py:358-390 forms them
    # (lower-case + NFKD, language tag first).  zh: the Cangjie table (mtl_tokenizer.py:139-143) is a download
    # and absent here, so the Han characters go through the vocabulary as they are; the count is recorded as such.
    def split_generic(text, max_chunk, sep):
        sentences = text.split(sep)
        n_chunks = len(text) // max_chunk + 1
        approx = len(text) // n_chunks
        chunks, cur, cur_len = [], [], 0
        for s in sentences:
            s = " ".join(s.split()).strip()
            if cur_len + len(s) > approx:
                chunks.append(sep.join(cur))
                cur, cur_len = [s], len(s)
            else:
                cur.append(s)
                cur_len += len(s)
        if cur:
            chunks.append(sep.join(cur))
        return [c + sep if re.match(r"[a-zA-Z0-9]", c[-1]) else c for c in chunks if len(c) > 0]

    for tag, fname, size, sep in (("fr", "docs/benchmark-text-fr-1.txt", 400, ". "), ("zh", "docs/benchmark-text-zh-1.txt", 50, "\u3002")):
        with open(os.path.join(REF, fname)) as f:
            text = f.read()
        lines = [l.strip() for l in text.split("\n") if not l.startswith("#") and l.strip()]
        chunks = [c for l in lines for c in split_generic(l, size, sep)]
        lens, chars = [], []
        for c in chunks:
            s_ = "[START]" + text_utils.punc_norm(c) + "[STOP]"
            pre = normalize("NFKD", s_.lower()).replace("[start]", "[START]").replace("[stop]", "[STOP]")
            lens.append(len(mtl_tok.encode(("[%s]" % tag + pre).replace(" ", "[SPACE]")).ids))
            chars.append(sum(1 for ch in c if not ch.isspace()))
        tj["benchmark_text_%s_1" % tag] = {"n_chunks": len(chunks), "text_id_counts": lens, "nonspace_chars": chars,
                                            "cangjie": False}
        print("benchmark", tag, "chunks", len(chunks), "ids", min(lens), "-", max(lens))
    with open(os.path.join(HERE, "ref_text_fixture.json"), "w") as f:
        json.dump(tj, f, ensure_ascii=False, indent=0)
    print("wrote ref_text_fixture.json: chunks", len(chunks), "mean ids", sum(lens) / len(lens), "max", max(lens))


if __name__ == "__main__":
    main()
