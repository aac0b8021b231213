"""EgoT2-g (HOI) — drop-in mirrors of HOI/models/multitask/video_model_builder.py:56-383
(`TaskPromptTransformer`, `TaskTranslationPromptTransformer`, `TaskTranslationPromptTransformer6Task`): one
encoder-decoder over the PNR, OSCC, action-recognition (SlowFast) and, for the 6-task model, LTA backbones, with the
task named by a prompt token. The shared task-translation ENCODER (d=512, 8 heads,
3 layers, S=48 or 4n) and the short sequence decoder + vocabulary head (row F1, egot2_amd/decoder.py) run in
libegot2x.so. The frozen backbones (`pnr_model`, `oscc_model`, `recognition_model`, `lta_model`) are built in the
constructors where the reference builds them (video_model_builder.py:98-130, :284-289) from args.{pnr,oscc,action,lta}_cfg_file
through egot2_amd.backbones.make_hoi_backbone; empty entries build none (feature-level use).
Also: `TaskTranslationPromptTransformer2Task` (video_model_builder_2task.py:50-167) and
`TaskTranslationPromptTransformerActionTask` (video_model_builder_action.py:21-187)."""
from __future__ import annotations

import copy
import math

import torch
import torch.nn as nn

from .backbones import freeze_backbone_params, freeze_params, make_hoi_backbone
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
        self._build_backbones(args, oscc_no_temp_pool)

    def _build_backbones(self, args, oscc_no_temp_pool):
        """Reference :98-130: PNR / OSCC frozen, SlowFast with a dim-wide trainable head."""
        if getattr(args, "pnr_cfg_file", None):
            self.pnr_model = make_hoi_backbone("pnr", cfg_file=args.pnr_cfg_file)
            freeze_params(self.pnr_model)
        if getattr(args, "oscc_cfg_file", None):
            self.oscc_model = make_hoi_backbone("oscc", cfg_file=args.oscc_cfg_file, no_temp_pool=oscc_no_temp_pool)
            freeze_params(self.oscc_model)
        if getattr(args, "action_cfg_file", None):
            self.recognition_model = make_hoi_backbone("slowfast", cfg_file=args.action_cfg_file, num_classes=[self.dim],
                                                       with_head=True, loader="recognition")
            freeze_backbone_params(self.recognition_model)  # do not freeze head

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
        if getattr(args, "lta_cfg_file", None):      # reference :284-289
            self.lta_model = make_hoi_backbone("lta", cfg_file=args.lta_cfg_file, build_decoder=False)
            freeze_params(self.lta_model)

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


class TaskPromptTransformer2Task(TaskPromptTransformer):
    """HOI/models/multitask/video_model_builder_2task.py:50-122: the PNR + OSCC EgoT2-g (two task embeddings, no action stream)."""

    def __init__(self, args, vocab, oscc_no_temp_pool=True):
        nn.Module.__init__(self)
        self.args = args
        self.vocab = vocab
        self.dim = args.hidden_dim
        self.n_tasks = 2
        self.task_dict = {'pnr': 0, 'oscc': 1}
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
        self.fc = nn.Linear(self.dim, len(self.vocab))
        self.ln = nn.LayerNorm(self.dim)
        self.task_embed = nn.Parameter(torch.randn(1, self.n_tasks, self.dim), requires_grad=True)
        self.pos_embed = PositionalEncoding(self.dim, dropout=0.1, max_len=200)
        self.embedding = nn.Embedding(len(self.vocab), self.dim)
        self.seq_len = 5
        self.y_mask = self.get_tgt_mask(self.seq_len)
        self._init_parameters()
        self._build_backbones(args, oscc_no_temp_pool)

    def _build_backbones(self, args, oscc_no_temp_pool):
        """Reference :88-100: PNR and OSCC only."""
        if getattr(args, "pnr_cfg_file", None):
            self.pnr_model = make_hoi_backbone("pnr", cfg_file=args.pnr_cfg_file)
            freeze_params(self.pnr_model)
        if getattr(args, "oscc_cfg_file", None):
            self.oscc_model = make_hoi_backbone("oscc", cfg_file=args.oscc_cfg_file, no_temp_pool=oscc_no_temp_pool)
            freeze_params(self.oscc_model)


class TaskTranslationPromptTransformer2Task(TaskPromptTransformer2Task):
    """Reference :124-167."""

    def encode_features(self, feat_pnr, feat_oscc):
        return self._encode_segments([feat_pnr, feat_oscc], [self.proj_pnr, self.proj_oscc], [0, 1], [0, 0])

    def encode(self, video_pnr):
        video_oscc = video_pnr.copy()
        with torch.no_grad():
            feat_pnr = self.pnr_model(video_pnr, middle=True)
            feat_oscc = self.oscc_model(video_oscc, middle=True)
        return self.encode_features(feat_pnr, feat_oscc)

    def forward(self, video_pnr, target):
        return self.decode(target, self.encode(video_pnr)).permute(1, 2, 0)

    def predict(self, video_pnr, task):
        assert task in ['pnr', 'oscc']
        batch_size = video_pnr[0].shape[0]
        encoded_x = self.encode(video_pnr)
        y = (torch.ones((batch_size, 1)) * self.vocab[task]).type_as(video_pnr[0]).long()
        return self.decode(y, encoded_x)[0, :]


class TaskTranslationPromptTransformerActionTask(nn.Module, TranslatorMixin, DecoderMixin):
    """HOI/models/multitask/video_model_builder_action.py:21-187: action-recognition / LTA EgoT2-g over the SlowFast clip
    feature and the LTA forecasting features. 'lta' prompts: tokens = ln(cat(action, lta)) + learned `pe` (1, 4, d) (both
    streams must already be hidden_dim wide, as in the reference); other prompts: ONE token, task embedding 0 + position 0.
    `v_idx` / `n_idx` (vocabulary index maps of utils.multitask.build_vocab.vocab_idx_to_orig) may be passed in; they are
    only used by predict()."""

    def __init__(self, args, vocab, v_idx=None, n_idx=None):
        super().__init__()
        self.args = args
        self.vocab = vocab
        if v_idx is None or n_idx is None:
            try:
                import importlib
                v_idx, n_idx = importlib.import_module("utils.multitask.build_vocab").vocab_idx_to_orig()
            except ImportError:
                pass
        self.v_idx, self.n_idx = v_idx, n_idx
        self.dim = args.hidden_dim
        self.n_tasks = 2
        self.n_heads = args.num_heads
        self.dim_feedforward = getattr(args, "ff_dim", 2048)
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
        self.fc = nn.Linear(self.dim, len(self.vocab))
        self.ln = nn.LayerNorm(self.dim)
        self.task_embed = nn.Parameter(torch.randn(1, self.n_tasks, self.dim), requires_grad=True)
        self.pos_embed = PositionalEncoding(self.dim, dropout=self.dp_rate, max_len=200)
        self.pe = nn.Parameter(torch.randn(1, 4, self.dim), requires_grad=True)
        self.embedding = nn.Embedding(len(self.vocab), self.dim)
        self.seq_len = 200
        self.y_mask = self.get_tgt_mask(self.seq_len)
        self._init_parameters()
        self.k = 1
        if getattr(args, "lta_cfg_file", None):     # reference :58-73
            import importlib
            cfg = importlib.import_module("utils.lta.parser").load_config_from_file(args.lta_cfg_file)
            self.k = cfg.FORECASTING.NUM_SEQUENCES_TO_PREDICT
            self.action_model = make_hoi_backbone("slowfast", cfg=cfg, num_classes=[self.dim], with_head=True,
                                                  ckpt=cfg.CHECKPOINT_FILE_PATH_AR, loader="lta")
            freeze_backbone_params(self.action_model)
            lta_cfg = copy.deepcopy(cfg)
            lta_cfg.FORECASTING.NUM_ACTIONS_TO_PREDICT = 20
            self.lta_model = make_hoi_backbone("lta", cfg=lta_cfg, build_decoder=True, ckpt=cfg.CHECKPOINT_FILE_PATH_LTA)
            freeze_params(self.lta_model)

    _init_parameters = TaskPromptTransformer._init_parameters
    get_tgt_mask = TaskPromptTransformer.get_tgt_mask
    decode = TaskPromptTransformer.decode

    def encode_clips(self, model, x):
        assert isinstance(x, list) and len(x) >= 1
        return torch.stack([model([pathway[:, i] for pathway in x]) for i in range(x[0].shape[1])], dim=1)

    def encode_features(self, task, feat_action, feat_lta=None):
        """'lta' in task: feat_action, feat_lta (B, n, d) with 2 n == 4 -> memory (2n, B, d); else feat_action (B, 1, d) -> (1, B, d)."""
        if 'lta' in task:
            feats = [feat_action, feat_lta]
            segs, off = [], 0
            for f in feats:
                segs.append(SegmentSpec(T=f.shape[1], d_in=f.shape[2], has_proj=False, add_row=None, pos_row0=off))
                off += f.shape[1]
            if off != self.pe.shape[1]:
                raise ValueError(f"token count {off} != {self.pe.shape[1]} learned positions")
            x = self._egx_encode(feats, segs, encoder=self.transformer_encoder, ln=self.ln, projs=[None, None],
                                 task_embed=None, pos_table=self.pe[0], p_drop=self.dp_rate)
        else:
            segs = [SegmentSpec(T=feat_action.shape[1], d_in=feat_action.shape[2], has_proj=False, add_row=0, pos_row0=0)]
            x = self._egx_encode([feat_action], segs, encoder=self.transformer_encoder, ln=self.ln, projs=[None],
                                 task_embed=self.task_embed, pos_table=self.pos_embed.pe, p_drop=self.dp_rate,
                                 p_pos=self.pos_embed.dropout.p)
        return x.permute(1, 0, 2)

    def encode(self, video, task):
        if 'lta' in task:  # only use tokens produced by action models
            feat_action = self.encode_clips(self.action_model, video)  # (bs, num_input, d)
            feat_lta = self.lta_model(video, None, middle=True).transpose(0, 1)  # (bs, num_input, d)
            return self.encode_features(task, feat_action.contiguous(), feat_lta.contiguous())
        feat_action = self.action_model(video).unsqueeze(1)  # (bs, 1, d)
        return self.encode_features(task, feat_action)

    def forward(self, video, target, task):
        assert task in ['action_verb', 'action_noun', 'lta_verb', 'lta_noun']
        return self.decode(target, self.encode(video, task)).permute(1, 2, 0)

    def predict(self, video, task):
        assert task in ['action', 'lta']
        encoded_x = self.encode(video, task)
        batch_size = encoded_x.shape[1]
        y_verb = (torch.ones((batch_size, 1)) * self.vocab[task + '_verb']).type_as(video[0]).long()
        preds_verb = self.decode(y_verb, encoded_x)[0, :, self.v_idx]
        y_noun = (torch.ones((batch_size, 1)) * self.vocab[task + '_noun']).type_as(video[0]).long()
        preds_noun = self.decode(y_noun, encoded_x)[0, :, self.n_idx]
        if task == 'lta':
            preds_verb = preds_verb.unsqueeze(dim=1)
            preds_noun = preds_noun.unsqueeze(dim=1)
        return [preds_verb, preds_noun]

    def generate(self, x):
        from torch.distributions.categorical import Categorical
        results = []
        for head_x in self.predict(x, 'lta'):
            if self.k > 1:
                preds_dist = Categorical(logits=head_x)
                preds = [preds_dist.sample() for _ in range(self.k)]
            elif self.k == 1:
                preds = [head_x.argmax(2)]
            results.append(torch.stack(preds, dim=1))
        return results
