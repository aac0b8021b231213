"""EgoT2-s translators for the TTM task — drop-in mirrors of
HHI/models/ttm/model_taskspecific.py:154-245 (`TaskFusionMFTransformer2Task`, `TaskFusionMFTransformer3Task`).

Same constructor (`Class(args)` reading args.{lam,ttm,asd}_checkpoint, nofreeze, hidden_dim, num_heads, dropout,
num_layers), same forward signatures, same parameter/buffer names and shapes. The arithmetic between the frozen
backbones' features and the logits runs in libegot2x.so (HIP, gfx950).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import functional as F_egx
from .backbones import freeze_params, make_backbone
from .functional import SegmentSpec
from .registry import make_registry
from .translator import PositionalEncoding, TranslatorMixin

MODEL_REGISTRY = make_registry("MODEL")


def build_model(args):
    """HHI/models/ttm/build.py:17-20."""
    return MODEL_REGISTRY.get(args.model)(args)


class TaskFusion3Task(nn.Module):
    """HHI/models/ttm/model_taskspecific.py:17-35 (backbone attach + freeze protocol, including the reference's
    behaviour of freezing `ttm_model` unless `nofreeze`)."""

    def __init__(self, lam_ckpt=None, ttm_ckpt=None, asd_ckpt=None, nofreeze=False):
        super().__init__()
        if lam_ckpt:
            self.lam_model = make_backbone("lam", lam_ckpt)
            freeze_params(self.lam_model)
        if ttm_ckpt:
            self.ttm_model = make_backbone("ttm", ttm_ckpt)
        if asd_ckpt:
            self.asd_model = make_backbone("asd", asd_ckpt)
            freeze_params(self.asd_model)
        if not nofreeze:
            print('Freezing task-specific models')
            freeze_params(self.ttm_model)

    def forward(self, video, video_asd, audio, audio_asd):
        raise NotImplementedError


class _TTMTranslator(TaskFusion3Task, TranslatorMixin):
    n_tasks = 0

    def _build(self, args):
        self.dim = args.hidden_dim
        self.n_heads = args.num_heads
        self.dp_rate = args.dropout
        self.num_layers = args.num_layers

    def _finish(self):
        self.task_embed = nn.Parameter(torch.randn(1, self.n_tasks, self.dim), requires_grad=True)
        self.pos_embed = PositionalEncoding(self.dim, dropout=0.1)
        # parameter container only (keys/shapes/init identical to the reference); never called
        self.transformer_encoder = nn.TransformerEncoder(
            encoder_layer=nn.TransformerEncoderLayer(d_model=self.dim, nhead=self.n_heads, dropout=self.dp_rate),
            num_layers=self.num_layers
        )
        self.ln = nn.LayerNorm(self.dim)
        self.linear_head = nn.Sequential(
            nn.LayerNorm(self.dim),
            nn.Linear(self.dim, 2)
        )

    def _tokens(self, feats, projs, task_ids, with_head=False, ce=None):
        segs = [SegmentSpec(T=f.shape[1], d_in=f.shape[2], has_proj=True, add_row=k, pos_row0=0)
                for f, k in zip(feats, task_ids)]
        head = (self.linear_head[0], self.linear_head[1]) if with_head else None
        return self._egx_encode(feats, segs, encoder=self.transformer_encoder, ln=self.ln, projs=projs,
                                task_embed=self.task_embed, pos_table=self.pos_embed.pe,
                                p_drop=self.dp_rate, p_pos=self.pos_embed.dropout.p, head=head, ce=ce)

    def _head(self, tokens):
        ln, fc = self.linear_head[0], self.linear_head[1]
        return F_egx.pool_head(tokens, ln.weight, ln.bias, fc.weight, fc.bias, ln.eps)


@MODEL_REGISTRY.register()
class TaskFusionMFTransformer2Task(_TTMTranslator):
    """Task Translation for 2 tasks: LAM and TTM (reference :154-194)."""

    def __init__(self, args):
        super().__init__(args.lam_checkpoint, args.ttm_checkpoint, None, args.nofreeze)
        self.n_tasks = 2
        self._build(args)
        self.proj_lam = nn.Linear(256, self.dim)
        self.proj_ttm = nn.Linear(256, self.dim)
        self._finish()

    def forward_features(self, ttm_out, lam_out, target=None, class_weight=None):
        """ttm_out, lam_out: (B, T, 256) backbone features -> (B, 2) logits. Token order ttm, lam (task ids 0, 1).
        target (B,) int64 [, class_weight (2,)]: -> (logits, loss) with loss = nn.CrossEntropyLoss(weight=class_weight)(logits, target)
        evaluated inside the forward (HHI/tasks/ttm/video_task_2loader.py:21-22,34; one launch less per step)."""
        return self._tokens([ttm_out, lam_out], [self.proj_ttm, self.proj_lam], [0, 1], with_head=True,
                            ce=None if target is None else (target, class_weight))

    def forward(self, video, audio):
        lam_out = self.lam_model(video, middle=True)  # (bs, T, 256)
        ttm_out = self.ttm_model(video, audio, middle=True)
        return self.forward_features(ttm_out, lam_out)


@MODEL_REGISTRY.register()
class TaskFusionMFTransformer3Task(_TTMTranslator):
    """Task Translation for 3 HHI tasks: LAM, TTM, ASD (reference :197-245)."""

    def __init__(self, args):
        super().__init__(args.lam_checkpoint, args.ttm_checkpoint, args.asd_checkpoint, args.nofreeze)
        self.n_tasks = 3
        self._build(args)
        self.proj_lam = nn.Linear(256, self.dim)
        self.proj_ttm = nn.Linear(256, self.dim)
        self.proj_asd = nn.Linear(256, self.dim)
        self._finish()

    def forward_features(self, ttm_out, lam_out, asd_out, target=None, class_weight=None):
        """(B, T, 256) features of the three backbones -> (B, 2) logits. Token order ttm, lam, asd = task ids 0, 1, 2.
        target (B,) int64 [, class_weight (2,)]: -> (logits, loss) with loss = nn.CrossEntropyLoss(weight=class_weight)(logits, target)
        evaluated inside the forward (HHI/tasks/ttm/video_task_2loader.py:21-22,34; one launch less per step)."""
        return self._tokens([ttm_out, lam_out, asd_out], [self.proj_ttm, self.proj_lam, self.proj_asd], [0, 1, 2],
                            with_head=True, ce=None if target is None else (target, class_weight))

    def forward(self, video, video_asd, audio, audio_asd):
        N, D, H, W = video_asd.shape
        audioEmbed = self.asd_model.forward_audio_frontend(audio_asd)
        visualEmbed = self.asd_model.forward_visual_frontend(video_asd)
        audioEmbed, visualEmbed = self.asd_model.forward_cross_attention(audioEmbed, visualEmbed)
        outsAV = self.asd_model.forward_audio_visual_backend(audioEmbed, visualEmbed)
        asd_out = outsAV.view(N, D, -1)  # (bs, T, 256)
        lam_out = self.lam_model(video, middle=True)
        ttm_out = self.ttm_model(video, audio, middle=True)
        return self.forward_features(ttm_out, lam_out, asd_out)
