"""HOI EgoT2-s translators for action recognition — drop-in mirrors of
HOI/models/lta/lta_models_transfer.py:96-137 (`TaskFusionMFTransformer3Task`: SlowFast slow/fast + PNR + OSCC tokens,
8 + 8 + 16 + 16 = 48) and :170-235 (`TaskFusionMFTransformer2TaskAR`: SlowFast slow/fast + LTA clip features,
8 + 8 + n = 18). Both share ONE LayerNorm between the token preparation and the two classification heads
(`linear_head{1,2} = Sequential(self.ln, Linear)`), use learned positions and no task embedding
(real recipe HOI/configs/recognition/ts_ar.yaml:47-55: d = 128, 8 heads, 3 layers, classes [115, 478])."""
from __future__ import annotations

import torch
import torch.nn as nn

from . import functional as F_egx
from .backbones import cfg_get, freeze_backbone_params, freeze_params, make_hoi_backbone
from .functional import SegmentSpec
from .registry import make_registry
from .translator import TranslatorMixin

MODEL_REGISTRY = make_registry("MODEL")


class _ARTranslator(nn.Module, TranslatorMixin):
    def _build_translator(self, cfg, sequence_len):
        num_cls1, num_cls2 = cfg.MODEL.NUM_CLASSES
        self.sequence_len = sequence_len
        self.num_heads = cfg.MODEL.TRANSLATION_HEADS
        self.num_layers = cfg.MODEL.TRANSLATION_LAYERS
        self.feature_dim = cfg.MODEL.TRANSLATION_INPUT_FEATURES
        self.dp_rate = cfg.MODEL.TRANSLATION_DROPOUT
        return num_cls1, num_cls2

    def _finish(self, num_cls1, num_cls2):
        self.pe = nn.Parameter(torch.randn(1, self.sequence_len, self.feature_dim), requires_grad=True)
        self.transformer = nn.TransformerEncoder(   # parameter container only
            encoder_layer=nn.TransformerEncoderLayer(d_model=self.feature_dim, nhead=self.num_heads,
                                                     dropout=self.dp_rate, batch_first=True),
            num_layers=self.num_layers)
        self.ln = nn.LayerNorm(self.feature_dim)
        self.linear_head1 = nn.Sequential(self.ln, nn.Linear(self.feature_dim, num_cls1))
        self.linear_head2 = nn.Sequential(self.ln, nn.Linear(self.feature_dim, num_cls2))

    def _translate(self, feats, projs):
        segs, off = [], 0
        for f in feats:
            segs.append(SegmentSpec(T=f.shape[1], d_in=f.shape[2], has_proj=True, add_row=None, pos_row0=off))
            off += f.shape[1]
        if off != self.sequence_len:
            raise ValueError(f"token count {off} != sequence_len {self.sequence_len}")
        tokens = self._egx_encode(feats, segs, encoder=self.transformer, ln=self.ln, projs=projs, task_embed=None,
                                  pos_table=self.pe[0], p_drop=self.dp_rate)
        y = F_egx.pool_head(tokens, self.ln.weight, self.ln.bias, None, None, self.ln.eps)   # LN(mean_s tokens), shared ln
        fc1, fc2 = self.linear_head1[1], self.linear_head2[1]
        return [F_egx.linear(y, fc1.weight, fc1.bias, self.egx_compute), F_egx.linear(y, fc2.weight, fc2.bias, self.egx_compute)]


@MODEL_REGISTRY.register()
class TaskFusionMFTransformer3Task(_ARTranslator):
    """Action recognition from SlowFast + PNR + OSCC features (reference :96-137)."""

    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        c1, c2 = self._build_translator(cfg, 48)
        self.proj1 = nn.Linear(8192, self.feature_dim)
        self.proj2 = nn.Linear(8192, self.feature_dim)
        self.proj3_slow = nn.Linear(2048, self.feature_dim)
        self.proj3_fast = nn.Linear(256, self.feature_dim)
        self.avg_pool_slow = nn.AdaptiveAvgPool3d((None, 1, 1))
        self.avg_pool_fast = nn.AdaptiveAvgPool3d((8, 1, 1))
        self._finish(c1, c2)
        # frozen backbones where the reference builds them (TaskFusion3Task.__init__, video_model_transfer_3task.py:23-58)
        from .hoi_pnr import build_task_backbones
        build_task_backbones(self, cfg, cfg_get(cfg, "PRETRAIN.PNR_CFG"), cfg_get(cfg, "PRETRAIN.OSCC_CFG"),
                             cfg_get(cfg, "PRETRAIN.ACTION_CFG"), oscc_no_temp_pool=True, action_with_head=False)

    def forward_features(self, action_feat_slow, action_feat_fast, pnr_feat, oscc_feat):
        """(B,8,2048), (B,8,256), (B,16,8192), (B,16,8192) -> [(B, n_verbs), (B, n_nouns)]; token order slow, fast,
        pnr, oscc as in the reference's torch.cat."""
        return self._translate([action_feat_slow, action_feat_fast, pnr_feat, oscc_feat],
                               [self.proj3_slow, self.proj3_fast, self.proj1, self.proj2])

    def forward(self, x_action, x_pnr):
        x_oscc = x_pnr.copy()
        pnr_feat = self.pnr_model(x_pnr, middle=True)
        oscc_feat = self.oscc_model(x_oscc, middle=True)
        x_action_list = self.recognition_model(x_action, middle=True)
        slow = self.avg_pool_slow(x_action_list[0]).squeeze(-1).squeeze(-1).permute(0, 2, 1)
        fast = self.avg_pool_fast(x_action_list[1]).squeeze(-1).squeeze(-1).permute(0, 2, 1)
        return self.forward_features(slow.contiguous(), fast.contiguous(), pnr_feat, oscc_feat)


@MODEL_REGISTRY.register()
class TaskFusionMFTransformer2TaskAR(_ARTranslator):
    """Action recognition from SlowFast + LTA clip features (reference :170-235)."""

    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        self.num_input = cfg.FORECASTING.NUM_INPUT_CLIPS
        self.input_offset = cfg.FORECASTING.INPUT_OFFSET
        c1, c2 = self._build_translator(cfg, 18)
        self.proj_lta = nn.Linear(2048, self.feature_dim)
        self.proj_slow = nn.Linear(2048, self.feature_dim)
        self.proj_fast = nn.Linear(256, self.feature_dim)
        self.avg_pool_slow = nn.AdaptiveAvgPool3d((None, 1, 1))
        self.avg_pool_fast = nn.AdaptiveAvgPool3d((8, 1, 1))
        self._finish(c1, c2)
        self._init_parameters()          # xavier on every matrix, before any backbone is attached (reference :204,222-225)
        if cfg_get(cfg, "PRETRAIN.ACTION_CFG"):     # reference :206-212
            self.action_model = make_hoi_backbone("slowfast", cfg_file=cfg.PRETRAIN.ACTION_CFG, num_classes=[self.feature_dim],
                                                  with_head=False, loader="lta")
            freeze_backbone_params(self.action_model)
        if cfg_get(cfg, "PRETRAIN.LTA_CFG"):        # reference :214-217
            self.lta_model = make_hoi_backbone("lta", cfg_file=cfg.PRETRAIN.LTA_CFG, build_decoder=False)
            freeze_params(self.lta_model)

    def _init_parameters(self):
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)

    def forward_features(self, action_feat_slow, action_feat_fast, feat_lta):
        """(B,8,2048), (B,8,256), (B,n,2048) with 16 + n = 18 -> [(B, n_verbs), (B, n_nouns)]."""
        return self._translate([action_feat_slow, action_feat_fast, feat_lta], [self.proj_slow, self.proj_fast, self.proj_lta])

    def forward(self, x):
        x1 = x.copy()
        x_action = [x1[0][:, -1, ...], x1[1][:, -1, ...]]
        x_lta = [x[0][:, 0:self.num_input, ...], x[1][:, 0:self.num_input, ...]]
        with torch.no_grad():
            x_action_list = self.action_model(x_action, middle=True)
            feat_lta = self.lta_model(x_lta, middle=True).transpose(0, 1)
        slow = self.avg_pool_slow(x_action_list[0]).squeeze(-1).squeeze(-1).permute(0, 2, 1)
        fast = self.avg_pool_fast(x_action_list[1]).squeeze(-1).squeeze(-1).permute(0, 2, 1)
        return self.forward_features(slow.contiguous(), fast.contiguous(), feat_lta.contiguous())
