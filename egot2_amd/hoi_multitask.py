"""EgoT2-g (HOI) — drop-in mirrors of HOI/models/multitask/video_model_builder.py:56-383
(`TaskPromptTransformer`, `TaskTranslationPromptTransformer`, `TaskTranslationPromptTransformer6Task`): one
encoder-decoder over the PNR, OSCC, action-recognition (SlowFast) and, for the 6-task model, LTA backbones, with the
task named by a prompt token. The shared task-translation ENCODER (d=512, 8 heads,
3 layers, S=48 or 4n) and the short sequence decoder + vocabulary head (row F1, egot2_amd/decoder.py) run in
libegot2x.so. The frozen backbones are attached by the host code (`pnr_model`, `oscc_model`,
`recognition_model`, `lta_model`), see INTEGRATION.md."""
from __future__ import annotations

import copy
import math

import torch
import torch.nn as nn

from .functional import SegmentSpec
from .hhi_multitask import CustomDecoderLayer
from .decoder import DecoderMixin
from .translator import PositionalEncoding, TranslatorMixin


class TaskPromptTransformer(nn.Module, TranslatorMixin, DecoderMixin):
    """Reference :56-216 (single-task prompts 'pnr' / 'oscc' / 'action')."""

    def __init__(self, args, vocab, oscc_no_temp_pool=True):
        super().__init__()
        self.args = args
        self.vocab = vocab
        self.dim = args.hidden_dim
        self.n_tasks = 3
        self.task_dict = {'pnr': 0, 'oscc': 1, 'action': 2}
        self.n_heads = args.num_heads
        self.num_layers = args.num_layers
        self.dp_rate = args.dropout
        self.transformer_encoder = nn.TransformerEncoder(   # parameter container only
            encoder_layer=nn.TransformerEncoderLayer(d_model=self.dim, nhead=self.n_heads, dropout=self.dp_rate),
            num_layers=self.num_layers
        )
        self.transformer_decoder = nn.TransformerDecoder(
            decoder_layer=CustomDecoderLayer(d_model=self.dim, nhead=self.n_heads, dropout=self.dp_rate),
            num_layers=self.num_layers
        )
        self.proj_pnr = nn.Linear(8192, self.dim)
        self.proj_oscc = nn.Linear(8192, self.dim)
        self.proj_action_slow = nn.Linear(2048, self.dim)
        self.proj_action_fast = nn.Linear(256, self.dim)
        self.avg_pool_slow = nn.AdaptiveAvgPool3d((None, 1, 1))
        self.avg_pool_fast = nn.AdaptiveAvgPool3d((8, 1, 1))
        self.fc = nn.Linear(self.dim, len(self.vocab))
        self.ln = nn.LayerNorm(self.dim)
        self.task_embed = nn.Parameter(torch.randn(1, self.n_tasks, self.dim), requires_grad=True)
        self.pos_embed = PositionalEncoding(self.dim, dropout=0.1, max_len=200)
        self.embedding = nn.Embedding(len(self.vocab), self.dim)
        self.seq_len = 5
        self.y_mask = self.get_tgt_mask(self.seq_len)   # plain attribute, not a buffer (as in the reference)
        self._init_parameters()

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
    def _encode_segments(self, feats, projs, task_ids, pos0):
        segs = [SegmentSpec(T=f.shape[1], d_in=f.shape[2], has_proj=p is not None, add_row=k, pos_row0=o)
                for f, p, k, o in zip(feats, projs, task_ids, pos0)]
        x = self._egx_encode(feats, segs, encoder=self.transformer_encoder, ln=self.ln, projs=projs,
                             task_embed=self.task_embed, pos_table=self.pos_embed.pe,
                             p_drop=self.dp_rate, p_pos=self.pos_embed.dropout.p)   # (B, S, d)
        return x.permute(1, 0, 2)                                                  # decoder memory (S, B, d)

    def _pool_action(self, x_action_list):
        slow = self.avg_pool_slow(x_action_list[0]).squeeze(-1).squeeze(-1).permute(0, 2, 1)
        fast = self.avg_pool_fast(x_action_list[1]).squeeze(-1).squeeze(-1).permute(0, 2, 1)
        return slow.contiguous(), fast.contiguous()

    def encode_task_features(self, task, feat=None, slow_feat=None, fast_feat=None):
        """Single-task memory (reference forward :160-180): 'pnr' / 'oscc' take (B,16,8192); 'action' takes the pooled
        SlowFast pathways (B,8,2048), (B,8,256), which share task id 2 and ONE position run 0..15."""
        if task == 'pnr':
            return self._encode_segments([feat], [self.proj_pnr], [0], [0])
        if task == 'oscc':
            return self._encode_segments([feat], [self.proj_oscc], [1], [0])
        return self._encode_segments([slow_feat, fast_feat], [self.proj_action_slow, self.proj_action_fast], [2, 2],
                                     [0, slow_feat.shape[1]])

    def _backbone_features(self, video, task):
        with torch.no_grad():
            if task == 'pnr':
                return dict(feat=self.pnr_model(video, middle=True))
            if task == 'oscc':
                return dict(feat=self.oscc_model(video, middle=True))
            slow, fast = self._pool_action(self.recognition_model(video, middle=True))
            return dict(slow_feat=slow, fast_feat=fast)

    # ---- decoder (HIP; row F1) ---------------------------------------------------------------------------
    def decode(self, y, encoded_x):
        """(B, sy) tokens + (S, B, d) memory -> (sy, B, |V|); on the GPU this is the HIP decoder (egot2_amd/decoder.py)."""
        return self._egx_decode(y, encoded_x, embedding=self.embedding, pos_embed=self.pos_embed,
                                decoder=self.transformer_decoder, fc=self.fc, n_heads=self.n_heads, p_drop=self.dp_rate)

    def forward(self, video, target, task):
        assert task in ['pnr', 'oscc', 'action']
        encoded_x = self.encode_task_features(task, **self._backbone_features(video, task))
        return self.decode(target, encoded_x).permute(1, 2, 0)

    def predict(self, video, task):
        assert task in ['pnr', 'oscc']
        batch_size = video[0].shape[0]
        with torch.no_grad():
            encoded_x = self.encode_task_features(task, **self._backbone_features(video, task))
        y = (torch.ones((batch_size, 1)) * self.vocab[task]).type_as(video[0]).long()
        return self.decode(y, encoded_x)[0, :]

    def _greedy(self, encoded_x, like, batch_size, seq_len=3):
        output_tokens = (torch.ones((batch_size, seq_len))).type_as(like).long()
        output_tokens[:, 0] = self.vocab['action']
        for sy in range(1, seq_len):
            output = torch.argmax(self.decode(output_tokens[:, :sy], encoded_x), dim=-1)
            output_tokens[:, sy] = output[-1, :]
        return output_tokens[:, 1:]

    def predict_ac(self, video):
        with torch.no_grad():
            encoded_x = self.encode_task_features('action', **self._backbone_features(video, 'action'))
        return self._greedy(encoded_x, video[0], video[0].shape[0])


class TaskTranslationPromptTransformer(TaskPromptTransformer):
    """Reference :219-275: all three backbones feed every prompt; 16 + 16 + (8 + 8) = 48 tokens."""

    def encode_features(self, feat_pnr, feat_oscc, slow_feat, fast_feat):
        return self._encode_segments([feat_pnr, feat_oscc, slow_feat, fast_feat],
                                     [self.proj_pnr, self.proj_oscc, self.proj_action_slow, self.proj_action_fast],
                                     [0, 1, 2, 2], [0, 0, 0, slow_feat.shape[1]])

    def encode(self, video_pnr, video_ac):
        video_oscc = video_pnr.copy()
        with torch.no_grad():
            feat_pnr = self.pnr_model(video_pnr, middle=True)
            feat_oscc = self.oscc_model(video_oscc, middle=True)
            slow, fast = self._pool_action(self.recognition_model(video_ac, middle=True))
        return self.encode_features(feat_pnr, feat_oscc, slow, fast)

    def forward(self, video_pnr, video_ac, target):
        return self.decode(target, self.encode(video_pnr, video_ac)).permute(1, 2, 0)

    def predict(self, video_pnr, video_ac, task):
        assert task in ['pnr', 'oscc', 'action_verb', 'action_noun']
        batch_size = video_pnr[0].shape[0]
        encoded_x = self.encode(video_pnr, video_ac)
        y = (torch.ones((batch_size, 1)) * self.vocab[task]).type_as(video_pnr[0]).long()
        output = self.decode(y, encoded_x)
        if 'action' in task:
            output = torch.argmax(output, dim=-1)
        return output[0, :]

    def predict_ac(self, video_pnr, video_ac):
        return self._greedy(self.encode(video_pnr, video_ac), video_pnr[0], video_pnr[0].shape[0])


class TaskTranslationPromptTransformer6Task(TaskPromptTransformer):
    """Reference :278-383: four backbones (PNR, OSCC, action recognition, LTA) and six target tasks; the 'lta' prompts
    use per-clip features (n clips each, S = 4n), every other prompt the 48-token layout."""

    def __init__(self, args, vocab):
        super().__init__(args, vocab)
        self.task_embed = nn.Parameter(torch.randn(1, 4, self.dim), requires_grad=True)
        self.proj_lta = nn.Linear(2048, self.dim)

    def encode_clips(self, model, x):
        assert isinstance(x, list) and len(x) >= 1
        return torch.stack([model([pathway[:, i] for pathway in x]) for i in range(x[0].shape[1])], dim=1)

    def encode_clips_pnr(self, model, x):
        return torch.stack([model([x[:, i, ...]], middle=True).mean(dim=1) for i in range(x.shape[1])], dim=1)

    def encode_features(self, task, feat_pnr, feat_oscc, a, b):
        """'lta' in task: a = feat_action (B, n, d) used as is (no projection), b = feat_lta (B, n, 2048);
        otherwise a, b = pooled SlowFast slow (B, 8, 2048) / fast (B, 8, 256)."""
        if 'lta' in task:
            return self._encode_segments([feat_pnr, feat_oscc, a, b], [self.proj_pnr, self.proj_oscc, None, self.proj_lta],
                                         [0, 1, 2, 3], [0, 0, 0, 0])
        return self._encode_segments([feat_pnr, feat_oscc, a, b],
                                     [self.proj_pnr, self.proj_oscc, self.proj_action_slow, self.proj_action_fast],
                                     [0, 1, 2, 2], [0, 0, 0, a.shape[1]])

    def encode(self, video_pnr, video_ac, task):
        with torch.no_grad():
            if 'lta' in task:
                video_oscc = copy.deepcopy(video_pnr)
                feat_pnr = self.encode_clips_pnr(self.pnr_model, video_pnr)
                feat_oscc = self.encode_clips_pnr(self.oscc_model, video_oscc)
                a = self.encode_clips(self.recognition_model, video_ac)
                b = self.lta_model(video_ac, None, middle=True).transpose(0, 1)
            else:
                video_oscc = video_pnr.copy()
                feat_pnr = self.pnr_model(video_pnr, middle=True)
                feat_oscc = self.oscc_model(video_oscc, middle=True)
                a, b = self._pool_action(self.recognition_model(video_ac, middle=True))
        return self.encode_features(task, feat_pnr.contiguous(), feat_oscc.contiguous(), a.contiguous(), b.contiguous())

    def forward(self, video_pnr, video_ac, target, task):
        return self.decode(target, self.encode(video_pnr, video_ac, task)).permute(1, 2, 0)

    def predict(self, video_pnr, video_ac, task, predict_verb_only=False, predict_noun_only=False):
        assert task in ['pnr', 'oscc', 'action', 'lta']
        encoded_x = self.encode(video_pnr, video_ac, task)
        batch_size = encoded_x.shape[1]
        if task in ['action', 'lta']:
            if not predict_noun_only:
                y_verb = (torch.ones((batch_size, 1)) * self.vocab[task + '_verb']).type_as(video_ac[0]).long()
                output_verb = self.decode(y_verb, encoded_x)
            if predict_verb_only:
                return
            y_noun = (torch.ones((batch_size, 1)) * self.vocab[task + '_noun']).type_as(video_ac[0]).long()
            output_noun = self.decode(y_noun, encoded_x)
            if predict_noun_only:
                return
            pred_verb = torch.argmax(output_verb, dim=-1)
            pred_noun = torch.argmax(output_noun, dim=-1)
            return torch.stack((pred_verb[0, :], pred_noun[0, :]), dim=1)
        y = (torch.ones((batch_size, 1)) * self.vocab[task]).type_as(video_pnr[0]).long()
        return self.decode(y, encoded_x)[0, :]
