"""HOI EgoT2-s translator for PNR / OSCC — drop-in mirror of
HOI/models/pnr/video_model_transfer_3task.py:212-258 (`TaskFusionMFTransformer3TaskDropout`): 16 + 16 + 8 + 8 = 48
tokens from the PNR, OSCC and SlowFast (slow / fast pathway) backbones, feature dropout before the shared
LayerNorm, learned positions, d_ff = 2 d, 8 heads, and a head whose first element IS the shared `ln`."""
from __future__ import annotations

import torch
import torch.nn as nn

from . import functional as F_egx
from .functional import SegmentSpec
from .registry import make_registry
from .translator import TranslatorMixin

MODEL_REGISTRY = make_registry("MODEL")


@MODEL_REGISTRY.register()
class TaskFusionMFTransformer3TaskDropout(nn.Module, TranslatorMixin):
    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        self.num_classes = 16 if "keyframe_localization" in cfg.DATA.TASK else 2
        self.unsqueeze_dim = 1 if "keyframe_localization" in cfg.DATA.TASK else 2
        self.sequence_len = 48
        self.feature_dim = cfg.MODEL.TRANSLATION_INPUT_FEATURES
        self.num_layers = cfg.MODEL.TRANSLATION_LAYERS
        self.proj1 = nn.Linear(8192, self.feature_dim)
        self.proj2 = nn.Linear(8192, self.feature_dim)
        self.proj3_slow = nn.Linear(2048, self.feature_dim)
        self.proj3_fast = nn.Linear(256, self.feature_dim)
        self.avg_pool_slow = nn.AdaptiveAvgPool3d((None, 1, 1))
        self.avg_pool_fast = nn.AdaptiveAvgPool3d((8, 1, 1))
        self.pe = nn.Parameter(torch.randn(1, self.sequence_len, self.feature_dim), requires_grad=True)
        self.ln = nn.LayerNorm(self.feature_dim)
        self.dp = nn.Dropout(cfg.MODEL.FEAT_DROPOUT_RATE)
        self.transformer = nn.TransformerEncoder(   # parameter container only
            encoder_layer=nn.TransformerEncoderLayer(d_model=self.feature_dim, nhead=8,
                                                     dropout=cfg.MODEL.TRANSFORMER_DROPOUT_RATE,
                                                     dim_feedforward=self.feature_dim * 2, batch_first=True),
            num_layers=self.num_layers)
        self.linear_head = nn.Sequential(self.ln, nn.Linear(self.feature_dim, self.num_classes))

    def forward_features(self, pnr_feat, oscc_feat, action_feat_slow, action_feat_fast):
        """(B,16,8192), (B,16,8192), (B,8,2048), (B,8,256) -> (B, 1|., num_classes) as the reference returns it."""
        feats = [pnr_feat, oscc_feat, action_feat_slow, action_feat_fast]
        projs = [self.proj1, self.proj2, self.proj3_slow, self.proj3_fast]
        segs, off = [], 0
        for f in feats:
            segs.append(SegmentSpec(T=f.shape[1], d_in=f.shape[2], has_proj=True, add_row=None, pos_row0=off))
            off += f.shape[1]
        assert off == self.sequence_len
        tokens = self._egx_encode(feats, segs, encoder=self.transformer, ln=self.ln, projs=projs, task_embed=None,
                                  pos_table=self.pe[0], p_drop=self.transformer.layers[0].dropout.p, p_feat=self.dp.p)
        fc = self.linear_head[1]
        out = F_egx.pool_head(tokens, self.ln.weight, self.ln.bias, fc.weight, fc.bias, self.ln.eps)
        return out.unsqueeze(self.unsqueeze_dim)

    def forward(self, x1, x2):
        with torch.no_grad():
            pnr_feat = self.pnr_model(x1, middle=True)
            oscc_feat = self.oscc_model(x1.copy(), middle=True)
            x_action_list = self.recognition_model(x2, middle=True)
            slow = self.avg_pool_slow(x_action_list[0]).squeeze(-1).squeeze(-1).permute(0, 2, 1)
            fast = self.avg_pool_fast(x_action_list[1]).squeeze(-1).squeeze(-1).permute(0, 2, 1)
        return self.forward_features(pnr_feat, oscc_feat, slow, fast)
