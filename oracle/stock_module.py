"""ORACLE / CPU BASELINE — test + bench infrastructure only (never imported by egot2_amd).

The reference translator's math as the stock torch.nn modules the reference itself instantiates
(HHI/models/ttm/model_taskspecific.py:197-245: three Linear(256, d), shared LayerNorm, task embedding, sinusoidal
PositionalEncoding with Dropout(0.1), nn.TransformerEncoder(nn.TransformerEncoderLayer(d, h, dropout=p), L),
token mean, LayerNorm + Linear(d, 2)), fed with backbone features. It is the reference class minus the backbone
attributes: its state_dict keys equal the reference's, and tests/test_oracle_golden.py checks it against the
golden fixtures generated from the imported reference. bench.py times it on the host cores as `cpu_baseline`
(kind "port"); /root/reference does not exist on the GPU box.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn


class PositionalEncoding(nn.Module):
    def __init__(self, d_model, dropout=0.1, max_len=1000):
        super().__init__()
        self.dropout = nn.Dropout(p=dropout)
        pe = torch.zeros(max_len, d_model)
        position = torch.arange(0, max_len, dtype=torch.float).unsqueeze(1)
        div_term = torch.exp(torch.arange(0, d_model, 2).float() * (-math.log(10000.0) / d_model))
        pe[:, 0::2] = torch.sin(position * div_term)
        pe[:, 1::2] = torch.cos(position * div_term)
        self.register_buffer('pe', pe.unsqueeze(0).transpose(0, 1))

    def forward(self, x):
        return self.dropout(x + self.pe[:x.size(0), :])


class StockTTMTranslator(nn.Module):
    """K in {2, 3}: token order ttm, lam[, asd] with task ids 0, 1[, 2]."""

    def __init__(self, n_tasks=3, dim=128, n_heads=4, dropout=0.5, num_layers=1):
        super().__init__()
        self.n_tasks = n_tasks
        self.proj_lam = nn.Linear(256, dim)
        self.proj_ttm = nn.Linear(256, dim)
        if n_tasks == 3:
            self.proj_asd = nn.Linear(256, dim)
        self.task_embed = nn.Parameter(torch.randn(1, n_tasks, dim), requires_grad=True)
        self.pos_embed = PositionalEncoding(dim, dropout=0.1)
        self.transformer_encoder = nn.TransformerEncoder(
            encoder_layer=nn.TransformerEncoderLayer(d_model=dim, nhead=n_heads, dropout=dropout),
            num_layers=num_layers)
        self.ln = nn.LayerNorm(dim)
        self.linear_head = nn.Sequential(nn.LayerNorm(dim), nn.Linear(dim, 2))

    def encode_prepare(self, x, task_id):
        x = self.ln(x) + self.task_embed[:, task_id, :]
        return self.pos_embed(x.permute(1, 0, 2))

    def forward(self, ttm_out, lam_out, asd_out=None):
        xs = [self.encode_prepare(self.proj_ttm(ttm_out), 0), self.encode_prepare(self.proj_lam(lam_out), 1)]
        if self.n_tasks == 3:
            xs.append(self.encode_prepare(self.proj_asd(asd_out), 2))
        out = self.transformer_encoder(torch.cat(xs, dim=0)).mean(dim=0)
        return self.linear_head(out)


def _time_steps(step, max_steps, budget_s, warmup):
    import time
    for _ in range(warmup):
        step()
    ts = []
    t_all = time.perf_counter()
    while len(ts) < max_steps and (time.perf_counter() - t_all) < budget_s:
        t0 = time.perf_counter()
        step()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    return ts[len(ts) // 2], len(ts), time.perf_counter() - t_all


def time_cpu_baseline(B=256, T=15, n_tasks=3, dim=128, n_heads=4, num_layers=1, dropout=0.5, budget_s=40.0,
                      max_steps=20, threads=None, one_thread_steps=3):
    """fwd + weighted-CE + bwd of the stock module on the host cores (BASELINE.md §3 protocol: all cores, 3 warm-up + 20
    timed steps, median; bounded by `budget_s` seconds), plus a 1-thread figure from a bounded sample of
    `one_thread_steps` steps (a full 20-step protocol at one thread takes minutes and would not fit the default bench run)."""
    all_threads = threads or torch.get_num_threads()
    torch.set_num_threads(all_threads)
    torch.manual_seed(0)
    m = StockTTMTranslator(n_tasks, dim, n_heads, dropout, num_layers).train()
    g = torch.Generator().manual_seed(1234)
    feats = [torch.randn(B, T, 256, generator=g) for _ in range(n_tasks)]
    y = torch.randint(0, 2, (B,), generator=g)
    crit = nn.CrossEntropyLoss(weight=torch.tensor([0.266, 0.734]))

    def step():
        m.zero_grad(set_to_none=True)
        crit(m(*feats), y).backward()

    med, n, dt = _time_steps(step, max_steps, budget_s, warmup=3)
    desc = (f"stock torch.nn translator, B={B}, T={T}, K={n_tasks}, d={dim}, L={num_layers}, dropout={dropout} (+0.1 PE), fp32")
    out = {"value": B / med, "unit": "clips/s", "cores": all_threads, "kind": "port",
           "sample": f"median of {n} fwd+bwd steps after 3 warm-up steps, {desc}, {dt:.1f}s"}
    if one_thread_steps > 0:
        torch.set_num_threads(1)
        try:
            med1, n1, dt1 = _time_steps(step, one_thread_steps, budget_s, warmup=1)
        finally:
            torch.set_num_threads(all_threads)
        out["one_thread"] = {"value": B / med1, "unit": "clips/s", "cores": 1,
                             "sample": f"median of {n1} steps after 1 warm-up step (bounded sample), {dt1:.1f}s"}
    return out
