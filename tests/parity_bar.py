"""The stated floating-point bar of the engine-versus-oracle tests, in one place (used by the GPU tests, by the CPU test
of the goldens and by tests/golden/make_oracle_golden.py, which refuses to write a golden that is not decisive by it).

The engine and the oracle round every op output to bf16 at the same points but sum in different orders, so a logit
can differ by a few bf16 ulp.  ``logit_tol`` is the tolerance on the CFG-combined, penalised fp32 logits at magnitudes
below 8:  (1 + 2*cfg_scale) * 2^-5 * depth(L)

  * 2^-5 = 1 bf16 ulp of a head logit in [4, 8) (2 ulp in [2, 4));
  * the CFG combine c + s*(c - u) amplifies an error of c or u by up to 1 + 2s;
  * depth(L) = max(1, sqrt(L / 8)): flipped roundings add up like a random walk over the layers.  Measured worst
    errors against this bar are kept in profiles/r02_logit_error_by_depth.json (tools/logit_error_by_depth.py).

``tol_at`` scales it with the bf16 ulp of larger logits (the decisive checkpoint's top logits are 16..32).
A step is DECISIVE when the oracle's top-2 margin exceeds ``DECISIVE_FACTOR`` tolerances at the top logit's magnitude:
there both implementations must pick the same id.
"""
import math

import torch

DECISIVE_FACTOR = 4.0


def depth_factor(n_layers: int) -> float:
    return max(1.0, math.sqrt(n_layers / 8.0))


def logit_tol(cfg_scale: float, n_layers: int = 2) -> float:
    return (1 + 2 * cfg_scale) * 2 ** -5 * depth_factor(n_layers)


def tol_at(logits: torch.Tensor, tol: float) -> torch.Tensor:
    """Per-element tolerance: ``tol`` up to |logit| < 8, then doubling with every binade (the bf16 ulp)."""
    a = logits.detach().abs().float().clamp(min=1.0)
    a = torch.where(torch.isfinite(a), a, torch.ones_like(a))
    return tol * torch.exp2(torch.clamp(torch.floor(torch.log2(a)) - 2, min=0.0))


def decisive_margin(top_logit: float, tol: float) -> float:
    return DECISIVE_FACTOR * float(tol_at(torch.tensor(float(top_logit)), tol))
