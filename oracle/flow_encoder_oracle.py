"""CPU restatement of the token -> mu half of S3Gen's flow stage.  TEST INFRASTRUCTURE ONLY: imported by tests/ and by the
golden-vector scripts, never by the package.

What it restates (reference files under /root/reference/src/chatterbox_vllm/models/s3gen/):
  CausalMaskedDiffWithXvec.inference, flow.py:246-268 up to `h = self.encoder_proj(h)`: speaker projection, token embedding,
  UpsampleConformerEncoder.forward (transformer/upsample_encoder.py:256-299) built with the arguments of s3gen.py:58-73
  (10 pre-norm blocks of relative-position self-attention + Swish feed-forward, no convolution module, no macaron), its
  PreLookaheadLayer (:66-99) and Upsample1D (:36-63), LinearNoSubsampling (transformer/subsampling.py:69-113),
  EspnetRelPositionalEncoding (transformer/embedding.py:201-294), RelPositionMultiHeadedAttention
  (transformer/attention.py:199-330), ConformerEncoderLayer.forward (transformer/encoder_layer.py:191-235).
The flow-matching decoder that consumes mu is not restated: its modules need `diffusers` and `conformer`, absent here.

Written as plain functions over a dict of tensors, batch 1 (the reference asserts it, flow.py:241), full attention (the
encoder is built with static_chunk_size 0 and no dynamic chunks, so add_optional_chunk_mask returns the padding mask,
which is all ones for one utterance).  The relative shift is written as the index rule it implements.

Pinned by tests/golden/flow_encoder_fixture.npz (outputs of the reference's own classes on the package's seeded weights);
tests/test_flow_encoder_oracle.py holds this file to them."""
import math

import torch
import torch.nn.functional as F

D_MODEL = 512
HEADS = 8
D_K = 64
FFN = 2048
N_BLOCKS = 6
N_UP_BLOCKS = 4
VOCAB = 6561
MEL = 80
SPK = 192
PRE_LOOKAHEAD = 3


def rel_positional_encoding(n, dtype=torch.float32):
    """embedding.py:224-254 + :290-293 for a sequence of n positions: rows 0 .. 2n-2 hold relative positions n-1 .. -(n-1);
    even columns sin(pos * w), odd columns cos(pos * w), w_i = exp(2i * -(ln 10000 / 512)), all in fp32 as the reference."""
    position = torch.arange(0, n, dtype=torch.float32).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, D_MODEL, 2, dtype=torch.float32) * -(math.log(10000.0) / D_MODEL))
    pos = torch.zeros(n, D_MODEL)
    neg = torch.zeros(n, D_MODEL)
    pos[:, 0::2] = torch.sin(position * div_term)
    pos[:, 1::2] = torch.cos(position * div_term)
    neg[:, 0::2] = torch.sin(-1 * position * div_term)
    neg[:, 1::2] = torch.cos(-1 * position * div_term)
    return torch.cat([torch.flip(pos, [0]), neg[1:]], dim=0).to(dtype)            # [2n - 1, 512]


def layer_norm(x, w, b, eps):
    return F.layer_norm(x, (x.shape[-1],), w, b, eps)


def rel_attention(w, p, x, pos_emb):
    """attention.py:255-330 for one utterance: x [T, 512] (already normalised), pos_emb [2T - 1, 512] -> [T, 512].
    scores[h, i, j] = ((q_i + u_h) . k_j + (q_i + v_h) . p_{T - 1 - i + j}) / 8: the relative shift of :231-252 picks, for
    query i and key j, the positional row of relative position i - j, which sits at index T - 1 - (i - j)."""
    T = x.shape[0]
    q = (x @ w[p + "linear_q.weight"].t() + w[p + "linear_q.bias"]).view(T, HEADS, D_K)
    k = (x @ w[p + "linear_k.weight"].t() + w[p + "linear_k.bias"]).view(T, HEADS, D_K)
    v = (x @ w[p + "linear_v.weight"].t() + w[p + "linear_v.bias"]).view(T, HEADS, D_K)
    pe = (pos_emb @ w[p + "linear_pos.weight"].t()).view(2 * T - 1, HEADS, D_K)
    qu = (q + w[p + "pos_bias_u"]).transpose(0, 1)                                  # [H, T, 64]
    qv = (q + w[p + "pos_bias_v"]).transpose(0, 1)
    ac = qu @ k.permute(1, 2, 0)                                                    # [H, T, T]
    bd_raw = qv @ pe.permute(1, 2, 0)                                               # [H, T, 2T - 1]
    idx = (T - 1) - torch.arange(T).view(T, 1) + torch.arange(T).view(1, T)         # [T, T]
    bd = torch.gather(bd_raw, 2, idx.unsqueeze(0).expand(HEADS, T, T))
    attn = torch.softmax((ac + bd) / math.sqrt(D_K), dim=-1)
    out = (attn @ v.transpose(0, 1)).transpose(0, 1).reshape(T, D_MODEL)
    return out @ w[p + "linear_out.weight"].t() + w[p + "linear_out.bias"]


def conformer_block(w, p, x, pos_emb):
    """encoder_layer.py:191-235 with normalize_before, no macaron, no convolution module (ff_scale 1)."""
    x = x + rel_attention(w, p + "self_attn.", layer_norm(x, w[p + "norm_mha.weight"], w[p + "norm_mha.bias"], 1e-12), pos_emb)
    y = layer_norm(x, w[p + "norm_ff.weight"], w[p + "norm_ff.bias"], 1e-12)
    y = y @ w[p + "feed_forward.w_1.weight"].t() + w[p + "feed_forward.w_1.bias"]
    y = y * torch.sigmoid(y)
    return x + (y @ w[p + "feed_forward.w_2.weight"].t() + w[p + "feed_forward.w_2.bias"])


def embed(w, p, x):
    """subsampling.py:111-113: Linear, LayerNorm(eps 1e-5), then embedding.py:267-270: scale by sqrt(512)."""
    x = x @ w[p + "out.0.weight"].t() + w[p + "out.0.bias"]
    x = layer_norm(x, w[p + "out.1.weight"], w[p + "out.1.bias"], 1e-5)
    return x * math.sqrt(D_MODEL)


def encoder(w, x, dtype=torch.float32, taps=None):
    """UpsampleConformerEncoder.forward: x [T, 512] -> [2T, 512]."""
    T = x.shape[0]
    p = "encoder."
    x = embed(w, p + "embed.", x)
    pos_emb = rel_positional_encoding(T, dtype)
    # pre-lookahead (upsample_encoder.py:82-99): 3 zero frames on the right, conv k=4, leaky ReLU, 2 zero frames on the
    # left, conv k=3, residual
    y = x.t().unsqueeze(0)
    y = F.leaky_relu(F.conv1d(F.pad(y, (0, PRE_LOOKAHEAD)), w[p + "pre_lookahead_layer.conv1.weight"], w[p + "pre_lookahead_layer.conv1.bias"]))
    y = F.conv1d(F.pad(y, (2, 0)), w[p + "pre_lookahead_layer.conv2.weight"], w[p + "pre_lookahead_layer.conv2.bias"])
    x = y.squeeze(0).t() + x
    if taps is not None:
        taps["pre_lookahead"] = x
    for i in range(N_BLOCKS):
        x = conformer_block(w, f"{p}encoders.{i}.", x, pos_emb)
    if taps is not None:
        taps["blocks"] = x
    # Upsample1D (upsample_encoder.py:58-63): nearest x2, 4 zero frames on the left, conv k=5
    y = x.t().unsqueeze(0)
    y = F.interpolate(y, scale_factor=2.0, mode="nearest")
    y = F.conv1d(F.pad(y, (4, 0)), w[p + "up_layer.conv.weight"], w[p + "up_layer.conv.bias"])
    x = y.squeeze(0).t()
    x = embed(w, p + "up_embed.", x)
    pos_emb = rel_positional_encoding(2 * T, dtype)
    for i in range(N_UP_BLOCKS):
        x = conformer_block(w, f"{p}up_encoders.{i}.", x, pos_emb)
    return layer_norm(x, w[p + "after_norm.weight"], w[p + "after_norm.bias"], 1e-5)


def token_to_mu(w, tokens, dtype=torch.float32, taps=None):
    """flow.py:252-262 with finalize=True: token ids [T] (prompt and new tokens already concatenated) -> mu [2T, 80]."""
    w = {k: v.to(dtype) for k, v in w.items()}
    ids = torch.clamp(tokens.long(), min=0, max=VOCAB - 1)
    x = w["input_embedding.weight"][ids]
    h = encoder(w, x, dtype, taps)
    return h @ w["encoder_proj.weight"].t() + w["encoder_proj.bias"]


def speaker_projection(w, embedding, dtype=torch.float32):
    """flow.py:244-245: L2-normalise the x-vector [192], project to 80."""
    e = F.normalize(embedding.to(dtype).view(1, -1), dim=1)
    return (e @ w["spk_embed_affine_layer.weight"].to(dtype).t() + w["spk_embed_affine_layer.bias"].to(dtype)).view(-1)
