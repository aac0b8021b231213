"""EgoT2-s translator for the ASD task — drop-in mirror of HHI/models/asd/model_taskspecific.py:108-158
(`TaskFusionMFTransformer3Task`): token order asd, ttm, lam (task ids 2, 0, 1), output = the encoded ASD token
block as (B*T, d); the classifier lives outside the model (HHI/tasks/asd/loss.py:11-30) and reads `output_dim`."""
from __future__ import annotations

import torch
import torch.nn as nn

from .backbones import freeze_params, make_backbone
from .functional import SegmentSpec
from .registry import make_registry
from .translator import PositionalEncoding, TranslatorMixin

MODEL_REGISTRY = make_registry("MODEL")


def build_model(args):
    """HHI/models/asd/build.py"""
    return MODEL_REGISTRY.get(args.model)(args)


class TaskFusion3Task(nn.Module):
    """HHI/models/asd/model_taskspecific.py:40-55 (all three backbones frozen; no `nofreeze` flag here)."""

    def __init__(self, lam_ckpt=None, ttm_ckpt=None, asd_ckpt=None):
        super().__init__()
        if lam_ckpt:
            self.lam_model = make_backbone("lam", lam_ckpt)
            freeze_params(self.lam_model)
        if ttm_ckpt:
            self.ttm_model = make_backbone("ttm", ttm_ckpt)
            freeze_params(self.ttm_model)
        if asd_ckpt:
            self.asd_model = make_backbone("asd", asd_ckpt)
            freeze_params(self.asd_model)

    def forward(self, video, video_asd, audio, audio_asd):
        raise NotImplementedError


@MODEL_REGISTRY.register()
class TaskFusionMFTransformer3Task(TaskFusion3Task, TranslatorMixin):
    def __init__(self, args):
        super().__init__(args.lam_checkpoint, args.ttm_checkpoint, args.asd_checkpoint)
        self.n_tasks = 3
        self.dim = args.hidden_dim
        self.n_heads = args.num_heads
        self.dp_rate = args.dropout
        self.num_layers = args.num_layers
        self.proj_lam = nn.Linear(256, self.dim)
        self.proj_ttm = nn.Linear(256, self.dim)
        self.proj_asd = nn.Linear(256, self.dim)
        self.task_embed = nn.Parameter(torch.randn(1, self.n_tasks, self.dim), requires_grad=True)
        self.pos_embed = PositionalEncoding(self.dim, dropout=0.1)
        self.transformer_encoder = nn.TransformerEncoder(   # parameter container only
            encoder_layer=nn.TransformerEncoderLayer(d_model=self.dim, nhead=self.n_heads, dropout=self.dp_rate),
            num_layers=self.num_layers
        )
        self.ln = nn.LayerNorm(self.dim)
        self.linear_head = nn.Sequential(   # present (and unused) in the reference too: kept for state_dict parity
            nn.LayerNorm(self.dim),
            nn.Linear(self.dim, 2)
        )
        self.output_dim = self.dim

    def forward_features(self, ttm_out, lam_out, asd_out, lossav=None, labels=None):
        """(B, T, 256) features -> (B*T, d): tokens asd | ttm | lam, first T tokens returned.
        lossav (the task's lossAV module) + labels (B*T,): return lossAV.forward(tokens, labels) = (nloss, predScore, predLabel, correctNum)
        instead (HHI/tasks/asd/video_task_taskspecific.py:24,33), evaluated by the encoder's own launches where the per-clip kernels can."""
        feats = [asd_out, ttm_out, lam_out]
        segs = [SegmentSpec(T=f.shape[1], d_in=f.shape[2], has_proj=True, add_row=k, pos_row0=0)
                for f, k in zip(feats, (2, 0, 1))]
        tokens = self._egx_encode(feats, segs, encoder=self.transformer_encoder, ln=self.ln,
                                  projs=[self.proj_asd, self.proj_ttm, self.proj_lam], task_embed=self.task_embed,
                                  pos_table=self.pos_embed.pe, p_drop=self.dp_rate, p_pos=self.pos_embed.dropout.p,
                                  out_tokens=asd_out.shape[1],     # x[0:D] of model_taskspecific.py:156-158, without the copy
                                  token_ce=None if lossav is None else (lossav.FC.weight, lossav.FC.bias, labels, lossav.criterion.weight))
        if lossav is not None:
            nloss, _, predScore, predLabel, correctNum = tokens
            return nloss, predScore, predLabel, correctNum
        N, D = asd_out.shape[0], asd_out.shape[1]
        return tokens.reshape(N * D, -1)

    def forward(self, video, video_asd, audio, audio_asd):
        with torch.no_grad():
            N, D, H, W = video_asd.shape
            audioEmbed = self.asd_model.forward_audio_frontend(audio_asd)
            visualEmbed = self.asd_model.forward_visual_frontend(video_asd)
            audioEmbed, visualEmbed = self.asd_model.forward_cross_attention(audioEmbed, visualEmbed)
            outsAV = self.asd_model.forward_audio_visual_backend(audioEmbed, visualEmbed)  # (N*D, 256)
            asd_out = outsAV.view(N, D, -1)
            lam_out = self.lam_model(video, middle=True)
            ttm_out = self.ttm_model(video, audio, middle=True)
        return self.forward_features(ttm_out, lam_out, asd_out)


class lossAV(nn.Module):
    """Drop-in for the ASD task's classifier + criterion, HHI/tasks/asd/loss.py:11-30: `FC = Linear(dim, 2)` and
    `CrossEntropyLoss(weight=[1, 4])` with the same attribute names (checkpoint keys `FC.weight`, `FC.bias`,
    `criterion.weight`) and the same return values. With labels the whole forward (logits, loss, softmax scores, rounded
    labels, number of correct frames) is one launch and its backward one launch (`functional.linear_cross_entropy`); the
    stock modules take ~10 launches and cost 15 % of the ASD translator step."""

    def __init__(self, dim: int = 256):
        super().__init__()
        self.criterion = nn.CrossEntropyLoss(weight=torch.FloatTensor([1, 4]))
        self.FC = nn.Linear(dim, 2)

    def forward(self, x, labels=None):
        from . import functional as F_egx
        x = x.squeeze(1)
        if labels is None:      # inference: scores of class 1 as a numpy vector, as the reference returns them
            z = F_egx.linear(x, self.FC.weight, self.FC.bias, "f32")
            return z[:, 1].t().reshape(-1).detach().cpu().numpy()
        nloss, _, predScore, predLabel, correctNum = F_egx.linear_cross_entropy(x, self.FC.weight, self.FC.bias, labels, self.criterion.weight)
        return nloss, predScore, predLabel, correctNum
