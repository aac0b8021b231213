"""HOI EgoT2-s translator for long-term anticipation — drop-in mirror of
HOI/models/lta/lta_models_lta_transfer.py:257-377 (`TaskFusionMFTransformerLTA4Task`) and of the
`MultiTaskHead` it decodes with (HOI/models/lta/head_helper.py:218-290).

Constructor reads the same yacs fields (cfg.FORECASTING.NUM_INPUT_CLIPS, cfg.MODEL.TRANSLATION_{HEADS,LAYERS,
INPUT_FEATURES,DROPOUT}, cfg.MODEL.{NUM_CLASSES,DROPOUT_RATE,HEAD_ACT}, cfg.TEST.NO_ACT,
cfg.FORECASTING.NUM_ACTIONS_TO_PREDICT); parameter names match the reference state_dict (pe, proj_{pnr,oscc,lta},
transformer.layers.*, ln, head.projections.*). The four frozen backbones are attached by the host code
(pnr_model, oscc_model, action_model, lta_model) — see INTEGRATION.md."""
from __future__ import annotations

from functools import reduce

import torch
import torch.nn as nn
from torch.distributions.categorical import Categorical

from . import functional as F_egx
from .functional import SegmentSpec
from .registry import make_registry
from .translator import TranslatorMixin

MODEL_REGISTRY = make_registry("MODEL")


class MultiTaskHead(nn.Module):
    """One Linear per future action on the pooled clip feature (head_helper.py:218-290, dim_in=[d], pool_size=[None]:
    the adaptive average pool over a (B, d, 1, 1, 1) input is the identity). Projections run through the HIP GEMM."""

    def __init__(self, dim_in, num_classes, pool_size, dropout_rate=0.0, act_func="softmax", test_noact=False):
        super().__init__()
        assert len({len(pool_size), len(dim_in)}) == 1, "pathway dimensions are not consistent."
        self.test_noact = test_noact
        if dropout_rate > 0.0:
            self.dropout = nn.Dropout(dropout_rate)
        self.projections = nn.ModuleList([nn.Linear(sum(dim_in), n, bias=True) for n in num_classes])
        if act_func == "softmax":
            self.act = nn.Softmax(dim=-1)
        elif act_func == "sigmoid":
            self.act = nn.Sigmoid()
        else:
            raise NotImplementedError("{} is not supported as an activation" "function.".format(act_func))
        self.egx_compute = "f32"

    def forward(self, feat):
        """feat: (B, d) -> list of (B, n_classes)."""
        if hasattr(self, "dropout"):
            feat = self.dropout(feat)
        # ONE GEMM for all Z future-action heads: the 20 (593, d) projections are stacked row-wise (autograd splits the
        # gradient back), instead of 20 launches of a 256-row problem each
        sizes = [p.out_features for p in self.projections]
        W = torch.cat([p.weight for p in self.projections], dim=0)
        b = torch.cat([p.bias for p in self.projections], dim=0)
        x = list(F_egx.linear(feat, W, b, self.egx_compute).split(sizes, dim=-1))
        if not self.training and not self.test_noact:
            x = [self.act(x_i) for x_i in x]
        return x


@MODEL_REGISTRY.register()
class TaskFusionMFTransformerLTA4Task(nn.Module, TranslatorMixin):
    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        self.sequence_len = cfg.FORECASTING.NUM_INPUT_CLIPS * 4
        self.num_heads = cfg.MODEL.TRANSLATION_HEADS
        self.num_layers = cfg.MODEL.TRANSLATION_LAYERS
        self.feature_dim = cfg.MODEL.TRANSLATION_INPUT_FEATURES
        self.dp_rate = cfg.MODEL.TRANSLATION_DROPOUT
        self.pe = nn.Parameter(torch.randn(1, self.sequence_len, self.feature_dim), requires_grad=True)
        self.proj_pnr = nn.Linear(8192, self.feature_dim)
        self.proj_oscc = nn.Linear(8192, self.feature_dim)
        self.proj_lta = nn.Linear(2048, self.feature_dim)
        self.transformer = nn.TransformerEncoder(   # parameter container only
            encoder_layer=nn.TransformerEncoderLayer(d_model=self.feature_dim, nhead=self.num_heads,
                                                     dropout=self.dp_rate, batch_first=True),
            num_layers=self.num_layers)
        self.ln = nn.LayerNorm(self.feature_dim)
        self._init_parameters()
        head_classes = [reduce((lambda x, y: x + y), cfg.MODEL.NUM_CLASSES)] * self.cfg.FORECASTING.NUM_ACTIONS_TO_PREDICT
        self.head = MultiTaskHead(dim_in=[self.feature_dim], num_classes=head_classes, pool_size=[None],
                                  dropout_rate=cfg.MODEL.DROPOUT_RATE, act_func=cfg.MODEL.HEAD_ACT,
                                  test_noact=cfg.TEST.NO_ACT)

    def _init_parameters(self):
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)

    def set_compute(self, compute="f32", impl="auto"):
        super().set_compute(compute, impl)
        self.head.egx_compute = compute
        return self

    def decode(self, x):
        x = torch.stack(self.head(x), dim=1)  # (B, Z, #verbs + #nouns)
        return torch.split(x, self.cfg.MODEL.NUM_CLASSES, dim=-1)

    def forward_features(self, feat_pnr, feat_oscc, feat_action, feat_lta):
        """pnr/oscc (B, n, 8192), action (B, n, d), lta (B, n, 2048) -> [(B, Z, #verbs), (B, Z, #nouns)]."""
        feats = [feat_pnr, feat_oscc, feat_action, feat_lta]
        projs = [self.proj_pnr, self.proj_oscc, None, self.proj_lta]
        segs, off = [], 0
        for f, pj in zip(feats, projs):
            segs.append(SegmentSpec(T=f.shape[1], d_in=f.shape[2], has_proj=pj is not None, add_row=None, pos_row0=off))
            off += f.shape[1]
        assert off == self.sequence_len, f"token count {off} != sequence_len {self.sequence_len}"
        tokens = self._egx_encode(feats, segs, encoder=self.transformer, ln=self.ln, projs=projs, task_embed=None,
                                  pos_table=self.pe[0], p_drop=self.dp_rate)
        pooled = F_egx.pool_head(tokens)   # mean over tokens
        return self.decode(pooled)

    def encode_clips(self, model, x):
        assert isinstance(x, list) and len(x) >= 1
        return torch.stack([model([pathway[:, i] for pathway in x]) for i in range(x[0].shape[1])], dim=1)

    def encode_clips_pnr(self, model, x):
        return torch.stack([model([x[:, i, ...]], middle=True).mean(dim=1) for i in range(x.shape[1])], dim=1)

    def forward(self, x_lta, x_pnr):
        with torch.no_grad():
            feat_pnr = self.encode_clips_pnr(self.pnr_model, x_pnr)
            feat_oscc = self.encode_clips_pnr(self.oscc_model, x_pnr)
            feat_lta = self.lta_model(x_lta, None, middle=True).transpose(0, 1)
        feat_action = self.encode_clips(self.action_model, x_lta)   # its head is trainable in the reference
        return self.forward_features(feat_pnr, feat_oscc, feat_action, feat_lta)

    def generate(self, x_lta, x_pnr, k=1):
        x = self.forward(x_lta, x_pnr)
        results = []
        for head_x in x:
            if k > 1:
                preds_dist = Categorical(logits=head_x)
                preds = [preds_dist.sample() for _ in range(k)]
            elif k == 1:
                preds = [head_x.argmax(2)]
            results.append(torch.stack(preds, dim=1))
        return results
