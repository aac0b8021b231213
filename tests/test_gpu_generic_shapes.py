"""-m gpu: the shape-generic kernels against the oracle over the corners of their claimed domain (egx_config: d_model a
multiple of 4 up to 1024, head dims 16/32/64/96/128, any S, any d_ff multiple of 4): outputs and every gradient."""
import pytest
import torch
import torch.nn as nn

from egot2_amd.functional import SegmentSpec
from egot2_amd.translator import PositionalEncoding, TranslatorMixin
from oracle import translator_ref as tr

pytestmark = pytest.mark.gpu


class _Mini(nn.Module, TranslatorMixin):
    def __init__(self, d, h, d_ins, d_ff, L):
        super().__init__()
        self.h = h
        self.projs = nn.ModuleList([nn.Linear(k, d) if k else nn.Identity() for k in d_ins])
        self.ln = nn.LayerNorm(d)
        self.task_embed = nn.Parameter(torch.randn(1, len(d_ins), d))
        self.pos_embed = PositionalEncoding(d, dropout=0.0)
        self.enc = nn.TransformerEncoder(nn.TransformerEncoderLayer(d_model=d, nhead=h, dim_feedforward=d_ff, dropout=0.0), num_layers=L)

    def run(self, feats):
        segs = [SegmentSpec(T=f.shape[1], d_in=f.shape[2], has_proj=isinstance(p, nn.Linear), add_row=k, pos_row0=0)
                for k, (f, p) in enumerate(zip(feats, self.projs))]
        return self._egx_encode(feats, segs, encoder=self.enc, ln=self.ln,
                                projs=[p if isinstance(p, nn.Linear) else None for p in self.projs],
                                task_embed=self.task_embed, pos_table=self.pos_embed.pe, p_drop=0.0, p_pos=0.0)


def _oracle(sd, h, feats, has_proj):
    pe = sd["pos_embed.pe"][:, 0, :]
    xs = []
    for k, f in enumerate(feats):
        w = sd.get(f"projs.{k}.weight") if has_proj[k] else None
        b = sd.get(f"projs.{k}.bias") if has_proj[k] else None
        xs.append(tr.encode_prepare(f, w, b, sd["ln.weight"], sd["ln.bias"], sd["task_embed"][0, k], pe[:f.shape[1]]))
    x = torch.cat(xs, dim=1)
    return tr.encoder(x, sd, "enc.", tr.n_layers_of(sd, "enc."), h)


CASES = [
    # (d, h, [(T, d_in or 0 = identity)], d_ff, L, B)
    (64, 4, [(5, 20), (3, 0)], 100, 1, 3),              # head dim 16, odd widths, an identity segment
    (192, 2, [(7, 64)], 4, 2, 2),                       # head dim 96, tiny FFN
    (320, 5, [(130, 128)], 512, 1, 2),                  # head dim 64, S = 130 (chunked keys)
    (1024, 8, [(9, 256), (8, 1024)], 2048, 1, 2),       # the largest width, head dim 128
    (256, 8, [(1, 8)], 3000, 1, 5),                     # S = 1, head dim 32, d_ff not a power of two
    (128, 4, [(257, 128)], 256, 1, 1),                  # S = 257: beyond the fused kernels, odd length
]


@pytest.mark.parametrize("case", CASES, ids=[f"case{i}" for i in range(len(CASES))])
def test_generic_kernels_match_oracle(egx_lib, cuda, case):
    d, h, segs, d_ff, L, B = case
    torch.manual_seed(d + h + d_ff)
    m = _Mini(d, h, [k for _, k in segs], d_ff, L)
    with torch.no_grad():
        for p in m.parameters():
            if p.dim() == 1:
                p.add_(0.1 * torch.randn_like(p))
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    feats = [torch.randn(B, T, (k or d)) for T, k in segs]
    md = m.to(cuda).set_compute("f32", "generic").train()
    y = md.run([f.to(cuda) for f in feats])
    w = torch.linspace(-1, 1, y.numel()).view(y.shape)
    (y * w.to(cuda)).sum().backward()
    sd64 = {k: v.double().requires_grad_(v.is_floating_point() and not k.endswith(".pe")) for k, v in sd.items()}
    ref = _oracle(sd64, h, [f.double() for f in feats], [bool(k) for _, k in segs])
    (ref * w.double()).sum().backward()
    assert (y.detach().cpu().double() - ref.detach()).abs().max().item() < 1e-3 * max(1.0, ref.abs().max().item())
    for k, p in md.named_parameters():
        r = sd64[k].grad
        if r is None:
            continue
        assert p.grad is not None, k
        err = (p.grad.detach().cpu().double() - r).norm().item() / (r.norm().item() + 1e-9)
        assert err < 1e-2, f"{k}: {err}"


def test_unsupported_head_dim_is_an_error_not_a_crash(egx_lib, cuda):
    from egot2_amd import _lib
    m = _Mini(80, 2, [16], 64, 1).to(cuda).set_compute("f32", "generic").train()      # head dim 40
    with pytest.raises(_lib.EgxError):
        m.run([torch.randn(2, 4, 16, device=cuda)])
