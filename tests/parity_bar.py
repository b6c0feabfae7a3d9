"""The stated floating-point bar of the engine-versus-oracle tests, in one place (used by the GPU tests, by the CPU test
of the goldens and by tests/golden/make_oracle_golden.py, which refuses to write a golden that is not decisive by it).

The engine and the oracle round every op output to bf16 at the same points but sum in different orders, so a logit
can differ by a few bf16 ulp.  ``logit_tol`` is the tolerance on the CFG-combined, penalised fp32 logits at magnitudes
below 8:  (1 + 2*cfg_scale) * 2^-5 * depth(L)

  * 2^-5 = 1 bf16 ulp of a head logit in [4, 8) (2 ulp in [2, 4));
  * the CFG combine c + s*(c - u) amplifies an error of c or u by up to 1 + 2s;
  * depth(L) = max(1, sqrt(L / 8)): flipped roundings add up like a random walk over the layers.  Measured worst
    errors against this bar are kept in profiles/r02_logit_error_by_depth.json (tools/logit_error_by_depth.py).

``tol_at`` widens it per column by the larger of two factors: the bf16 ulp of a larger logit (doubling with every
binade from 8 up; the decisive checkpoint's top logits are 16..32), and the norm of the column's speech-head row
relative to the median row (a logit is a dot product of that row with the hidden state, so the few-ulp differences
of the hidden state weigh in proportion to the row's norm; 1 for every row of the flat checkpoint, up to 3.4 for
the rows the decisive checkpoint adds structure to).  Measured (profiles/r02_logit_error_by_depth.json): worst error
0.57-0.91 tolerances on the flat checkpoint at 2..30 layers and CFG 0.5 / 3.0.
A step is DECISIVE when the oracle's top-2 margin exceeds ``DECISIVE_FACTOR`` tolerances of its top-2 columns:
there both implementations must pick the same id.
"""
import math

import torch

DECISIVE_FACTOR = 3.0


def depth_factor(n_layers: int) -> float:
    return max(1.0, math.sqrt(n_layers / 8.0))


def logit_tol(cfg_scale: float, n_layers: int = 2) -> float:
    return (1 + 2 * cfg_scale) * 2 ** -5 * depth_factor(n_layers)


def head_row_scale(state_dict) -> torch.Tensor:
    """Norm of every speech-head row over the median norm, as a [1000 + vocab] vector in the sampler's column space
    (the first 1000 columns are the -inf padding: scale 1)."""
    n = state_dict["speech_head.weight"].float().norm(dim=1)
    return torch.cat([torch.ones(1000), (n / n.median()).clamp(min=1.0)])


def tol_at(logits: torch.Tensor, tol: float, row_scale=None) -> torch.Tensor:
    """Per-element tolerance: ``tol`` times the larger of the binade factor (1 up to |logit| < 8, then doubling with
    every binade, as the bf16 ulp does) and ``row_scale`` (same shape as ``logits``, or a number; None = 1)."""
    a = logits.detach().abs().float().clamp(min=1.0)
    a = torch.where(torch.isfinite(a), a, torch.ones_like(a))
    f = torch.exp2(torch.clamp(torch.floor(torch.log2(a)) - 2, min=0.0))
    if row_scale is not None:
        f = torch.maximum(f, torch.as_tensor(row_scale, dtype=torch.float32).expand_as(f))
    return tol * f


def decisive_margin(top_logit: float, tol: float, row_scale: float = 1.0) -> float:
    """Margin above which a step counts as decisive; ``row_scale`` = the larger head-row scale of the top-2 columns."""
    return DECISIVE_FACTOR * float(tol_at(torch.tensor(float(top_logit)), tol, float(row_scale)))


def step_is_decisive(logits: torch.Tensor, tol: float, row_scale=None) -> bool:
    v, i = torch.topk(logits, 2)
    sc = 1.0 if row_scale is None else float(torch.as_tensor(row_scale)[i].max())
    return (v[0] - v[1]).item() > decisive_margin(v[0].item(), tol, sc)
