"""The flow-encoder oracle (oracle/flow_encoder_oracle.py) against tests/golden/flow_encoder_fixture.npz: what the
reference's own CausalMaskedDiffWithXvec.inference hands to its flow-matching decoder (mu, speaker vector, condition) on
the package's seeded weights (tests/golden/make_flow_encoder_fixtures.py).  CPU only."""
import math
import os

import numpy as np
import pytest
import torch

from chatterbox_vllm_amd import flow_weights as FW
from oracle import flow_encoder_oracle as FO

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def fx():
    return np.load(os.path.join(ROOT, "tests", "golden", "flow_encoder_fixture.npz"))


@pytest.fixture(scope="module")
def w():
    return FW.synthetic_state_dict()


def case_tokens(fx, tag):
    return torch.cat([FW.synthetic_tokens(int(fx[f"{tag}_n_prompt"]), seed=3), FW.synthetic_tokens(int(fx[f"{tag}_n_new"]), seed=4)])


def test_inventory_counts():
    n = sum(math.prod(s) for _, s in FW.tensor_inventory())
    assert n == 6561 * 512 + 80 * 192 + 80 + 2 * (512 * 512 + 3 * 512) + 1024 + 512 * 512 * 12 + 3 * 512 \
        + 10 * (5 * 512 * 512 + 4 * 512 + 2 * 512 + 2 * 512 * 2048 + 2048 + 512 + 4 * 512) + 80 * 512 + 80


@pytest.mark.parametrize("tag", ["a", "b"])
def test_mu_matches_the_reference(fx, w, tag):
    mu = FO.token_to_mu(w, case_tokens(fx, tag))
    ref = torch.from_numpy(fx[f"{tag}_mu"])[0].t()
    assert mu.shape == ref.shape
    err = (mu - ref).abs().max().item()
    assert err <= 2e-5 * max(1.0, ref.abs().max().item()), err


def test_encoder_output_and_speaker_projection(fx, w):
    taps = {}
    tok = case_tokens(fx, "a")
    x = w["input_embedding.weight"][tok]
    h = FO.encoder(w, x, taps=taps)
    ref = torch.from_numpy(fx["a_encoder_out"])[0]
    assert (h - ref).abs().max().item() <= 2e-5 * ref.abs().max().item()
    for tag in ("a", "b"):
        s = FO.speaker_projection(w, torch.from_numpy(fx[f"{tag}_xvec"]))
        assert (s - torch.from_numpy(fx[f"{tag}_spks"])[0]).abs().max().item() <= 1e-6


def test_fp64_restatement_is_within_rounding(fx, w):
    mu64 = FO.token_to_mu({k: v.double() for k, v in w.items()}, case_tokens(fx, "b"), dtype=torch.float64)
    ref = torch.from_numpy(fx["b_mu"])[0].t()
    assert (mu64.float() - ref).abs().max().item() <= 5e-5
