"""HOI EgoT2-s translator for PNR / OSCC — drop-in mirror of
HOI/models/pnr/video_model_transfer_3task.py:212-258 (`TaskFusionMFTransformer3TaskDropout`): 16 + 16 + 8 + 8 = 48
tokens from the PNR, OSCC and SlowFast (slow / fast pathway) backbones, feature dropout before the shared
LayerNorm, learned positions, d_ff = 2 d, 8 heads, and a head whose first element IS the shared `ln`.

Also `TaskFusionMFTransformer` (HOI/models/pnr/video_model_transfer.py:44-67), the PNR/OSCC translator over the PRE-LN
`simple_vit.Transformer` (HOI/models/pnr/simple_vit.py:55-107): no token-prep LayerNorm, bias-free attention projections
with an inner width (heads x dim_head = 8 x 128) wider than the model (256), exact GELU. Its blocks are composed from the
library's single operations (MFMA GEMMs with the residual in the epilogue, LayerNorm, attention, GELU).

Frozen backbones are built in the constructors exactly where the reference builds them (`TaskFusion3Task.__init__`,
video_model_transfer_3task.py:23-58; `TaskFusion.__init__`, video_model_transfer.py:18-41) through
egot2_amd.backbones.make_hoi_backbone; empty config entries build none."""
from __future__ import annotations

import torch
import torch.nn as nn

from . import functional as F_egx
from .backbones import cfg_get, freeze_backbone_params, freeze_params, make_hoi_backbone
from .functional import SegmentSpec
from .registry import make_registry
from .translator import TranslatorMixin

MODEL_REGISTRY = make_registry("MODEL")


def build_task_backbones(model, cfg, cfg_pnr_file=None, cfg_oscc_file=None, cfg_recognition_file=None, oscc_no_temp_pool=False,
                         action_with_head=True):
    """`TaskFusion3Task.__init__` (video_model_transfer_3task.py:23-58): PNR / OSCC / SlowFast backbones from their config
    files, frozen (and put in eval mode) when cfg.PRETRAIN.{PNR,OSCC,ACTION}_FT is set."""
    model.cfg_pnr = None
    model.cfg_recognition = None
    if cfg_pnr_file:
        model.pnr_model = make_hoi_backbone("pnr", cfg_file=cfg_pnr_file)
        if cfg_get(cfg, "PRETRAIN.PNR_FT", True):
            model.pnr_model.eval()
            freeze_params(model.pnr_model)
    if cfg_oscc_file:
        model.oscc_model = make_hoi_backbone("oscc", cfg_file=cfg_oscc_file, no_temp_pool=oscc_no_temp_pool)
        if cfg_get(cfg, "PRETRAIN.OSCC_FT", True):
            model.oscc_model.eval()
            freeze_params(model.oscc_model)
    if cfg_recognition_file:
        model.recognition_model = make_hoi_backbone("slowfast", cfg_file=cfg_recognition_file,
                                                    num_classes=[cfg.MODEL.TRANSLATION_INPUT_FEATURES],
                                                    with_head=action_with_head, loader="recognition")
        if cfg_get(cfg, "PRETRAIN.ACTION_FT", True):
            model.recognition_model.eval()
            freeze_backbone_params(model.recognition_model)   # the head stays trainable


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
        build_task_backbones(self, cfg, cfg_get(cfg, "PRETRAIN.PNR_CFG"), cfg_get(cfg, "PRETRAIN.OSCC_CFG"),
                             cfg_get(cfg, "PRETRAIN.ACTION_CFG"), oscc_no_temp_pool=True, action_with_head=False)

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


# ---- pre-LN translator over simple_vit.Transformer ------------------------------------------------------------------------
class _VitAttention(nn.Module):
    """Parameter container with the names of simple_vit.Attention (:67-92): norm, to_qkv (no bias), to_out (no bias)."""

    def __init__(self, dim, heads=8, dim_head=64):
        super().__init__()
        inner_dim = dim_head * heads
        self.heads = heads
        self.scale = dim_head ** -0.5
        self.norm = nn.LayerNorm(dim)
        self.to_qkv = nn.Linear(dim, inner_dim * 3, bias=False)
        self.to_out = nn.Linear(inner_dim, dim, bias=False)


class _VitFeedForward(nn.Module):
    """simple_vit.FeedForward (:55-65): net = [LayerNorm, Linear, GELU, Linear]."""

    def __init__(self, dim, hidden_dim):
        super().__init__()
        self.net = nn.Sequential(nn.LayerNorm(dim), nn.Linear(dim, hidden_dim), nn.GELU(), nn.Linear(hidden_dim, dim))


class _VitTransformer(nn.Module):
    """simple_vit.Transformer (:94-107): layers[i] = ModuleList([Attention, FeedForward]); x = attn(x) + x; x = ff(x) + x."""

    def __init__(self, dim, depth, heads, dim_head, mlp_dim):
        super().__init__()
        self.layers = nn.ModuleList([nn.ModuleList([_VitAttention(dim, heads=heads, dim_head=dim_head), _VitFeedForward(dim, mlp_dim)])
                                     for _ in range(depth)])


@MODEL_REGISTRY.register()
class TaskFusionMFTransformer(nn.Module, TranslatorMixin):
    """mid fusion transformer (reference video_model_transfer.py:44-67): 16 + 16 tokens, d = 256, depth 3, 8 heads of 128."""

    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        self.num_classes = 16 if cfg.DATA.TASK == "keyframe_localization" else 2
        self.unsqueeze_dim = 1 if cfg.DATA.TASK == "keyframe_localization" else 2
        self.sequence_len = 32
        self.feature_dim = 256
        self.proj1 = nn.Linear(8192, self.feature_dim)
        self.proj2 = nn.Linear(8192, self.feature_dim)
        self.pe = nn.Parameter(torch.randn(1, self.sequence_len, self.feature_dim), requires_grad=True)
        self.transformer = _VitTransformer(dim=self.feature_dim, depth=3, heads=8, dim_head=128, mlp_dim=512)   # parameter container
        self.linear_head = nn.Sequential(nn.LayerNorm(self.feature_dim), nn.Linear(self.feature_dim, self.num_classes))
        build_task_backbones(self, cfg, cfg_get(cfg, "PRETRAIN.PNR_CFG"), cfg_get(cfg, "PRETRAIN.OSCC_CFG"), None,
                             oscc_no_temp_pool=True)

    def forward_features(self, pnr_feat, oscc_feat):
        """(B, 16, 8192) x 2 -> (B, 1 | ., num_classes)."""
        comp = self.egx_compute
        B, S, d = pnr_feat.shape[0], pnr_feat.shape[1] + oscc_feat.shape[1], self.feature_dim
        if S != self.sequence_len:
            raise ValueError(f"token count {S} != sequence_len {self.sequence_len}")
        feat = torch.cat((F_egx.linear(pnr_feat, self.proj1.weight, self.proj1.bias, comp),
                          F_egx.linear(oscc_feat, self.proj2.weight, self.proj2.bias, comp)), dim=1) + self.pe
        x = feat.reshape(B * S, d)
        for attn, ff in self.transformer.layers:
            h = F_egx.layer_norm_residual(x, None, attn.norm.weight, attn.norm.bias, attn.norm.eps)
            a = F_egx.attention(F_egx.linear(h, attn.to_qkv.weight, None, comp), B, S, attn.heads)
            x = F_egx.linear_residual(a, attn.to_out.weight, None, x, comp)
            ln, fc1, fc2 = ff.net[0], ff.net[1], ff.net[3]
            h = F_egx.layer_norm_residual(x, None, ln.weight, ln.bias, ln.eps)
            g = F_egx.gelu(F_egx.linear(h, fc1.weight, fc1.bias, comp))
            x = F_egx.linear_residual(g, fc2.weight, fc2.bias, x, comp)
        hl, fc = self.linear_head[0], self.linear_head[1]
        out = F_egx.pool_head(x.view(B, S, d), hl.weight, hl.bias, fc.weight, fc.bias, hl.eps)
        return out.unsqueeze(self.unsqueeze_dim)

    def forward(self, x):
        x2 = x.copy()
        pnr_feat = self.pnr_model(x, middle=True)  # (bs, 16, 8192)
        oscc_feat = self.oscc_model(x2, middle=True)  # (bs, 16, 8192)
        return self.forward_features(pnr_feat, oscc_feat)
