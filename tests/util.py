"""Shared helpers for the parity tests."""
from egot2_amd.synth import hhi_args  # noqa: F401  (re-exported for the tests)

import numpy as np
import torch


def seeded_feats(seed, shapes):
    """Features from numpy's PCG64 (stable across platforms), fp32."""
    rng = np.random.default_rng(seed)
    return [torch.from_numpy(rng.standard_normal(s, dtype=np.float32)) for s in shapes]


def seeded_state_dict(model, seed, scale=None):
    """Deterministic weights independent of torch's RNG and of state_dict ordering: every floating tensor except
    the sinusoid buffer is drawn from PCG64 seeded by (seed, crc32(key)) with a fan-in style scale; LayerNorm
    weights around 1."""
    import zlib
    sd = {}
    for k, v in model.state_dict().items():
        if k.endswith("pos_embed.pe") or not v.is_floating_point():
            sd[k] = v.clone()
            continue
        rng = np.random.default_rng([seed, zlib.crc32(k.encode())])
        a = rng.standard_normal(tuple(v.shape), dtype=np.float32)
        is_norm = ("norm" in k or k.startswith("ln.") or k.endswith("linear_head.0.weight") or k.endswith("linear_head.0.bias"))
        if is_norm and k.endswith("weight"):
            a = 1.0 + 0.1 * a
        elif k.endswith("bias"):
            a *= 0.1
        elif v.dim() >= 2 and "task_embed" not in k and k != "pe" and "embedding" not in k:
            a *= (1.0 / np.sqrt(v.shape[-1]))
        sd[k] = torch.from_numpy(a.astype(np.float32))
    return sd


def rel_err(a: torch.Tensor, b: torch.Tensor) -> float:
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def max_err(a: torch.Tensor, b: torch.Tensor) -> float:
    return (a.double().cpu() - b.double().cpu()).abs().max().item()
