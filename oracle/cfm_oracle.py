"""CPU restatement of S3Gen's flow-matching decoder (mu -> mel).  TEST INFRASTRUCTURE ONLY: imported by tests/, by
__graft_entry__.smoke() and by the CPU-baseline leg of the benchmarks, never by the package.

PARITY: pinned to outputs of the reference's own code, except for the inside of three third-party classes.
tests/golden/make_cfm_fixtures.py imports the reference's decoder.py, matcha/decoder.py, matcha/transformer.py and
flow_matching.py unmodified (the absent packages `conformer`, `diffusers`, `omegaconf` are stand-in modules) and records
their outputs on this repository's seeded weights -> tests/golden/cfm_fixture.npz; tests/test_cfm_oracle.py holds this file
to them:
  * SinusoidalPosEmb, TimestepEmbedding, CausalConv1d, CausalBlock1D, CausalResnetBlock1D, and solve_euler /
    CausalConditionalCFM.forward around a closed-form estimator: the reference's own arithmetic alone;
  * ConditionalDecoder.forward whole and the 10-step solve around it: the reference's own module tree and forward code
    with the three classes it takes from diffusers (`Attention`, `GELU`, `LoRACompatibleLinear`; the dependency is
    unpinned in pyproject.toml:25) replaced by labelled minimal stand-ins written from diffusers' published behaviour:
      Attention(query_dim=256, heads=8, dim_head=64, bias=False), self-attention, default processor:
          q, k, v = to_q(x), to_k(x), to_v(x) (no bias) split into 8 heads of 64;
          softmax(q k^T / sqrt(64) + mask) v; to_out[0] (Linear 512 -> 256 with bias); to_out[1] is dropout 0
      GELU(256, 1024): Linear then F.gelu (erf form);  LoRACompatibleLinear without adapters: nn.Linear.
STILL UNPINNED: the inside of those three classes (diffusers is not installed; the reference holds no output of them).

What it restates (reference files under /root/reference/src/chatterbox_vllm/models/s3gen/):
  ConditionalDecoder.forward, decoder.py:233-317 (one down stage, twelve mid stages, one up stage with the skip
  connection, final block and projection; all masks are ones for one utterance and static_chunk_size is 0, so
  add_optional_chunk_mask returns the padding mask and the attention bias is zero);
  CausalConv1d / CausalBlock1D / CausalResnetBlock1D, decoder.py:48-97 and matcha/decoder.py:46-61;
  SinusoidalPosEmb and TimestepEmbedding, matcha/decoder.py:14-29, 73-118;
  BasicTransformerBlock.forward, matcha/transformer.py:243-316 (norm1 -> attn1 -> residual, norm3 -> ff -> residual);
  CausalConditionalCFM.forward and ConditionalCFM.solve_euler, flow_matching.py:203-228, 82-133."""
import math

import torch
import torch.nn.functional as F

P = "decoder.estimator."
MEL = 80
CH = 256
HEADS = 8
HEAD_DIM = 64
N_BLOCKS = 4
N_MID = 12
CFG_RATE = 0.7


def sinusoidal_pos_emb(t, dim=320, scale=1000):
    """matcha/decoder.py:20-29: t [B] -> [B, dim]; the frequency table is computed in fp32 as the reference does."""
    half = dim // 2
    emb = math.log(10000) / (half - 1)
    emb = torch.exp(torch.arange(half).float() * -emb).to(t.dtype)
    emb = scale * t.unsqueeze(1) * emb.unsqueeze(0)
    return torch.cat((emb.sin(), emb.cos()), dim=-1)


def time_embedding(w, t):
    """decoder.py:251-252: SinusoidalPosEmb(320) then Linear, SiLU, Linear -> [B, 1024]."""
    e = sinusoidal_pos_emb(t)
    e = F.silu(e @ w[P + "time_mlp.linear_1.weight"].t() + w[P + "time_mlp.linear_1.bias"])
    return e @ w[P + "time_mlp.linear_2.weight"].t() + w[P + "time_mlp.linear_2.bias"]


def causal_conv(w, name, x):
    """decoder.py:94-97: two zero frames on the left, kernel 3."""
    return F.conv1d(F.pad(x, (2, 0)), w[name + ".weight"], w[name + ".bias"])


def causal_block(w, p, x):
    """decoder.py:48-61 with an all-ones mask: causal conv, LayerNorm over channels, Mish."""
    y = causal_conv(w, p + "block.0", x)
    y = F.layer_norm(y.transpose(1, 2), (CH,), w[p + "block.2.weight"], w[p + "block.2.bias"], 1e-5).transpose(1, 2)
    return F.mish(y)


def resnet_block(w, p, x, temb):
    """matcha/decoder.py:56-61."""
    h = causal_block(w, p + "block1.", x)
    h = h + (F.mish(temb) @ w[p + "mlp.1.weight"].t() + w[p + "mlp.1.bias"]).unsqueeze(-1)
    h = causal_block(w, p + "block2.", h)
    return h + F.conv1d(x, w[p + "res_conv.weight"], w[p + "res_conv.bias"])


def transformer_block(w, p, x):
    """matcha/transformer.py:243-316 for the configuration in use (plain LayerNorm, self-attention only, GELU
    feed-forward); x [B, T, 256]."""
    B, T, _ = x.shape
    n = F.layer_norm(x, (CH,), w[p + "norm1.weight"], w[p + "norm1.bias"], 1e-5)
    q = (n @ w[p + "attn1.to_q.weight"].t()).view(B, T, HEADS, HEAD_DIM).transpose(1, 2)
    k = (n @ w[p + "attn1.to_k.weight"].t()).view(B, T, HEADS, HEAD_DIM).transpose(1, 2)
    v = (n @ w[p + "attn1.to_v.weight"].t()).view(B, T, HEADS, HEAD_DIM).transpose(1, 2)
    a = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(HEAD_DIM), dim=-1)
    o = (a @ v).transpose(1, 2).reshape(B, T, HEADS * HEAD_DIM)
    x = x + (o @ w[p + "attn1.to_out.0.weight"].t() + w[p + "attn1.to_out.0.bias"])
    n = F.layer_norm(x, (CH,), w[p + "norm3.weight"], w[p + "norm3.bias"], 1e-5)
    f = F.gelu(n @ w[p + "ff.net.0.proj.weight"].t() + w[p + "ff.net.0.proj.bias"])
    return x + (f @ w[p + "ff.net.2.weight"].t() + w[p + "ff.net.2.bias"])


def stage(w, p, x, temb):
    x = resnet_block(w, p + "0.", x, temb)
    x = x.transpose(1, 2)
    for j in range(N_BLOCKS):
        x = transformer_block(w, f"{p}1.{j}.", x)
    return x.transpose(1, 2)


def estimator(w, x, mu, t, spks, cond, taps=None):
    """ConditionalDecoder.forward, decoder.py:233-317: x, mu, cond [B, 80, T], t [B], spks [B, 80] -> [B, 80, T]."""
    temb = time_embedding(w, t)
    T = x.shape[-1]
    x = torch.cat([x, mu, spks.unsqueeze(-1).expand(-1, -1, T), cond], dim=1)
    x = stage(w, P + "down_blocks.0.", x, temb)
    if taps is not None:
        taps["down"] = x
    skip = x
    x = causal_conv(w, P + "down_blocks.0.2", x)
    for i in range(N_MID):
        x = stage(w, f"{P}mid_blocks.{i}.", x, temb)
    if taps is not None:
        taps["mid"] = x
    x = stage(w, P + "up_blocks.0.", torch.cat([x, skip], dim=1), temb)
    x = causal_conv(w, P + "up_blocks.0.2", x)
    x = causal_block(w, P + "final_block.", x)
    return F.conv1d(x, w[P + "final_proj.weight"], w[P + "final_proj.bias"])


def solve_euler(w, z, t_span, mu, spks, cond, dtype=torch.float32, sol=None, estimator_fn=None):
    """flow_matching.py:82-133 for one utterance: row 0 of every estimator call is conditioned, row 1 sees zeros for mu,
    spks and cond; the two velocity estimates are combined with the guidance rate 0.7.  t and dt advance in the
    arithmetic of t_span (fp32 in the reference), whatever `dtype` the network runs in.  `estimator_fn(x, mu, t, spks,
    cond)` replaces the network (the fixture's closed-form estimator)."""
    w = {k: v.to(dtype) for k, v in w.items()} if estimator_fn is None else None
    x = z.to(dtype)
    t, dt = t_span[0], t_span[1] - t_span[0]
    n = len(t_span)
    for step in range(1, n):
        x_in = torch.cat([x, x], dim=0)
        mu_in = torch.cat([mu.to(dtype), torch.zeros_like(mu, dtype=dtype)], dim=0)
        sp_in = torch.cat([spks.to(dtype), torch.zeros_like(spks, dtype=dtype)], dim=0)
        c_in = torch.cat([cond.to(dtype), torch.zeros_like(cond, dtype=dtype)], dim=0)
        t_in = torch.stack([t, t]).to(dtype)
        d = estimator(w, x_in, mu_in, t_in, sp_in, c_in) if estimator_fn is None else estimator_fn(x_in, mu_in, t_in, sp_in, c_in)
        d = (1.0 + CFG_RATE) * d[0:1] - CFG_RATE * d[1:2]
        x = x + dt.to(dtype) * d
        t = t + dt
        if sol is not None:
            sol.append(x)
        if step < n - 1:
            dt = t_span[step + 1] - t
    return x


def cfm_forward(w, mu, spks, cond, noise, n_timesteps=10, temperature=1.0, dtype=torch.float32, estimator_fn=None):
    """CausalConditionalCFM.forward, flow_matching.py:203-228; `noise` stands for the leading columns of the module's
    `rand_noise` buffer."""
    z = noise[:, :, :mu.shape[2]] * temperature
    t_span = torch.linspace(0, 1, n_timesteps + 1, dtype=torch.float32)
    t_span = 1 - torch.cos(t_span * 0.5 * torch.pi)
    return solve_euler(w, z, t_span, mu, spks, cond, dtype, estimator_fn=estimator_fn)
