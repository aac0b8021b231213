"""EgoT2-g (HHI) — drop-in mirror of HHI/models/multitask/task_prompt_model.py:174-293
(`TaskTranslationPromptTransformer`). The shared task-translation ENCODER (the graded hot path)
runs in libegot2x.so; and so does the 2-token sequence decoder + vocabulary head (egot2_amd/decoder.py)."""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from .backbones import freeze_params, make_backbone
from .functional import SegmentSpec
from .decoder import DecoderMixin
from .translator import PositionalEncoding, TranslatorMixin


class CustomDecoderLayer(nn.TransformerDecoderLayer):
    """task_prompt_model.py:163-172 (need_weights=True only changes the discarded second return value); the extra
    is_causal argument is what torch >= 2 passes."""

    def __init__(self, d_model, nhead, dropout=0.1):
        super().__init__(d_model, nhead, dropout=dropout)

    def _mha_block(self, x, mem, attn_mask, key_padding_mask, is_causal=False):
        x = self.multihead_attn(x, mem, mem, attn_mask=attn_mask, key_padding_mask=key_padding_mask, need_weights=True)[0]
        return self.dropout2(x)


class TaskTranslationPromptTransformer(nn.Module, TranslatorMixin, DecoderMixin):
    def __init__(self, args, vocab):
        super().__init__()
        self.args = args
        self.vocab = vocab
        self.n_tasks = 3
        self.dim = args.hidden_dim
        self.n_heads = args.num_heads
        self.num_layers = args.num_layers
        self.dp_rate = args.dropout
        self.max_output_length = 500
        self.transformer_encoder = nn.TransformerEncoder(   # parameter container only
            encoder_layer=nn.TransformerEncoderLayer(d_model=self.dim, nhead=self.n_heads, dropout=self.dp_rate),
            num_layers=self.num_layers
        )
        self.transformer_decoder = nn.TransformerDecoder(
            decoder_layer=CustomDecoderLayer(d_model=self.dim, nhead=self.n_heads, dropout=self.dp_rate),
            num_layers=self.num_layers
        )
        self.ln = nn.LayerNorm(self.dim)
        self.task_embed = nn.Parameter(torch.randn(1, self.n_tasks, self.dim), requires_grad=True)
        self.pos_embed = PositionalEncoding(self.dim, dropout=0.1)
        self.embedding = nn.Embedding(len(self.vocab), self.dim)
        self.proj_lam = nn.Linear(256, self.dim)
        self.proj_ttm = nn.Linear(256, self.dim)
        self.proj_asd = nn.Linear(256, self.dim)
        self.fc = nn.Linear(self.dim, len(self.vocab))
        self.seq_len = 2
        self.y_mask = self.get_tgt_mask(self.seq_len)   # plain attribute, not a buffer (as in the reference)
        self._init_parameters()
        if getattr(args, "lam_checkpoint", None):
            self.lam_model = make_backbone("lam", args.lam_checkpoint)
            freeze_params(self.lam_model)
        if getattr(args, "ttm_checkpoint", None):
            self.ttm_model = make_backbone("ttm", args.ttm_checkpoint)
            freeze_params(self.ttm_model)
        if getattr(args, "asd_checkpoint", None):
            self.asd_model = make_backbone("asd", args.asd_checkpoint)
            freeze_params(self.asd_model)

    def _init_parameters(self):
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)

    def get_tgt_mask(self, size) -> torch.Tensor:
        mask = torch.tril(torch.ones(size, size) == 1).float()
        mask = mask.masked_fill(mask == 0, float('-inf'))
        mask = mask.masked_fill(mask == 1, float(0.0))
        return mask

    # ---- encoder (HIP) ---------------------------------------------------------------------------------
    def encode_features(self, task, lam_feat, ttm_feat=None, asd_feat=None):
        """Backbone features -> decoder memory in the reference layout: (S, B, d), or (3, B*T, d) for 'asd'."""
        if task == 'lam':
            feats, projs, ids = [lam_feat], [self.proj_lam], [0]
        else:
            feats, projs, ids = [lam_feat, ttm_feat, asd_feat], [self.proj_lam, self.proj_ttm, self.proj_asd], [0, 1, 2]
        segs = [SegmentSpec(T=f.shape[1], d_in=f.shape[2], has_proj=True, add_row=k, pos_row0=0) for f, k in zip(feats, ids)]
        x = self._egx_encode(feats, segs, encoder=self.transformer_encoder, ln=self.ln, projs=projs,
                             task_embed=self.task_embed, pos_table=self.pos_embed.pe,
                             p_drop=self.dp_rate, p_pos=self.pos_embed.dropout.p)   # (B, S, d)
        if task == 'asd':
            T = x.shape[1] // 3
            return torch.stack((x[:, 0:T].reshape(-1, self.dim), x[:, T:2 * T].reshape(-1, self.dim),
                                x[:, 2 * T:3 * T].reshape(-1, self.dim)), dim=0)
        return x.permute(1, 0, 2)

    def encode(self, video, video_asd, audio, audio_asd, task):
        with torch.no_grad():
            lam_feat = self.lam_model(video, middle=True)
            if task == 'lam':
                return self.encode_features(task, lam_feat)
            ttm_feat = self.ttm_model(video, audio, middle=True)
            N, D, H, W = video_asd.shape
            audioEmbed = self.asd_model.forward_audio_frontend(audio_asd)
            visualEmbed = self.asd_model.forward_visual_frontend(video_asd)
            audioEmbed, visualEmbed = self.asd_model.forward_cross_attention(audioEmbed, visualEmbed)
            outsAV = self.asd_model.forward_audio_visual_backend(audioEmbed, visualEmbed)
            asd_feat = outsAV.view(N, D, -1)
        return self.encode_features(task, lam_feat, ttm_feat, asd_feat)

    # ---- decoder (HIP; row F1) ---------------------------------------------------------------------------
    def decode(self, y, encoded_x):
        """(B, sy) tokens + (S, B, d) memory -> (sy, B, |V|); on the GPU this is the HIP decoder (egot2_amd/decoder.py)."""
        return self._egx_decode(y, encoded_x, embedding=self.embedding, pos_embed=self.pos_embed,
                                decoder=self.transformer_decoder, fc=self.fc, n_heads=self.n_heads, p_drop=self.dp_rate)

    def forward(self, video, video_asd, audio, audio_asd, target, task):
        assert task in ['lam', 'ttm', 'asd']
        encoded_x = self.encode(video, video_asd, audio, audio_asd, task)
        return self.decode(target, encoded_x).permute(1, 2, 0)

    def predict(self, video, video_asd, audio, audio_asd, task):
        assert task in ['lam', 'ttm', 'asd']
        batch_size = video.shape[0] * video.shape[1] if task == 'asd' else video.shape[0]
        encoded_x = self.encode(video, video_asd, audio, audio_asd, task)
        y = torch.ones((batch_size, 1)) * self.vocab[task]
        y = y.type_as(video).long()
        output = self.decode(y, encoded_x)
        return output[0, :, -2:]
