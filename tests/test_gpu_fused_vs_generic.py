"""-m gpu: the fused per-clip kernels against the shape-generic kernels (themselves pinned to the oracle) on shapes
the model classes never produce: ragged segment lengths, 1..4 segments, feature widths 128/384/512, d_ff from 128 to
4096, up to 4 layers, odd clip counts - outputs and every gradient, in fp32."""
import pytest
import torch
import torch.nn as nn

from egot2_amd.functional import SegmentSpec
from egot2_amd.translator import PositionalEncoding, TranslatorMixin

pytestmark = pytest.mark.gpu


class _Mini(nn.Module, TranslatorMixin):
    def __init__(self, d_ins, d_ff, L, use_task_embed=True):
        super().__init__()
        d = 128
        self.projs = nn.ModuleList([nn.Linear(k, d) for k in d_ins])
        self.ln = nn.LayerNorm(d)
        self.task_embed = nn.Parameter(torch.randn(1, len(d_ins), d)) if use_task_embed else None
        self.pos_embed = PositionalEncoding(d, dropout=0.0)
        self.enc = nn.TransformerEncoder(nn.TransformerEncoderLayer(d_model=d, nhead=4, dim_feedforward=d_ff, dropout=0.0), num_layers=L)
        self.head_ln, self.head_fc = nn.LayerNorm(d), nn.Linear(d, 3)

    def run(self, feats, with_head):
        segs = [SegmentSpec(T=f.shape[1], d_in=f.shape[2], has_proj=True, add_row=(k if self.task_embed is not None else None), pos_row0=0)
                for k, f in enumerate(feats)]
        return self._egx_encode(feats, segs, encoder=self.enc, ln=self.ln, projs=list(self.projs), task_embed=self.task_embed,
                                pos_table=self.pos_embed.pe, p_drop=0.0, p_pos=0.0,
                                head=(self.head_ln, self.head_fc) if with_head else None)


CASES = [
    # (B, [(T, d_in), ...], d_ff, L, head)
    (3, [(7, 256), (16, 128), (25, 384)], 2048, 1, True),      # S = 48 exactly, ragged, mixed widths
    (5, [(1, 128)], 128, 1, False),                            # single token, smallest FFN
    (2, [(10, 512), (3, 256), (9, 128), (20, 256)], 256, 2, True),   # four segments
    (7, [(15, 256), (15, 256)], 4096, 1, True),                # wide FFN
    (4, [(33, 256)], 1024, 4, False),                          # one long segment, four layers
    (1, [(16, 256), (16, 256), (16, 256)], 512, 3, True),
    (9, [(2, 128), (5, 128)], 2048, 2, False),                 # S = 7: two of the three token tiles are padding
]


@pytest.mark.parametrize("case", CASES, ids=[f"case{i}" for i in range(len(CASES))])
def test_fused_matches_generic(egx_lib, cuda, case):
    B, segs, d_ff, L, with_head = case
    torch.manual_seed(1234 + B + d_ff)
    m = _Mini([k for _, k in segs], d_ff, L).to(cuda).train()
    with torch.no_grad():
        for p in m.parameters():
            if p.dim() == 1:
                p.add_(0.1 * torch.randn_like(p))      # biases / LN weights away from their 0 / 1 defaults
    feats = [torch.randn(B, T, k, device=cuda) for T, k in segs]
    outs, grads = {}, {}
    for impl in ("generic", "fused"):
        m.set_compute("f32", impl)
        m.zero_grad(set_to_none=True)
        y = m.run(feats, with_head)
        w = torch.linspace(-1, 1, y.numel(), device=cuda).view_as(y)
        (y * w).sum().backward()
        outs[impl] = y.detach().clone()
        grads[impl] = {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}
    assert (outs["fused"] - outs["generic"]).abs().max().item() < 2e-4 * max(1.0, outs["generic"].abs().max().item())
    assert set(grads["fused"]) == set(grads["generic"])
    for n, g in grads["generic"].items():
        err = (grads["fused"][n] - g).norm().item() / (g.norm().item() + 1e-6)
        assert err < 5e-3, f"{n}: {err}"
