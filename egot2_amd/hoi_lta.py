"""HOI EgoT2-s translator for long-term anticipation — drop-in mirror of
HOI/models/lta/lta_models_lta_transfer.py:257-377 (`TaskFusionMFTransformerLTA4Task`) and of the
`MultiTaskHead` it decodes with (HOI/models/lta/head_helper.py:218-290).

Constructor reads the same yacs fields (cfg.FORECASTING.NUM_INPUT_CLIPS, cfg.MODEL.TRANSLATION_{HEADS,LAYERS,
INPUT_FEATURES,DROPOUT}, cfg.MODEL.{NUM_CLASSES,DROPOUT_RATE,HEAD_ACT}, cfg.TEST.NO_ACT,
cfg.FORECASTING.NUM_ACTIONS_TO_PREDICT); parameter names match the reference state_dict (pe, proj_{pnr,oscc,lta},
transformer.layers.*, ln, head.projections.*). The four frozen backbones are built in the constructor exactly where the
reference builds them (lta_models_lta_transfer.py:279-302), from cfg.PRETRAIN.{PNR,OSCC}_CFG and
cfg.CHECKPOINT_FILE_PATH_{AR,LTA}, through egot2_amd.backbones.make_hoi_backbone (reference classes when the HOI tree is
importable, or registered factories); a config without those entries builds none (feature-level use).
Also here: `TaskFusionMFTransformer2Task` (reference :429-526), the LTA translator over the action + LTA streams only."""
from __future__ import annotations

from functools import reduce

import torch
import torch.nn as nn
from torch.distributions.categorical import Categorical

from . import functional as F_egx
from .backbones import cfg_get, freeze_backbone_params, freeze_params, make_hoi_backbone
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

    def _stacked(self):
        """(Wst, bst): the projections' weights / biases as ONE row-wise stack that is their storage. Checked on every call (twenty pointer
        compares); when the parameters do not lie behind one another in one storage (fresh module, module.to(), load_state_dict(assign=True))
        the values are copied into a new stack and every `p.data` becomes a view of it — Parameter objects, state_dict keys and shapes stay what
        they were. FusedAdam's flat buffer, which lays the parameters out like their gradients, is such a storage and is used as it is."""
        ws, bs = [p.weight for p in self.projections], [p.bias for p in self.projections]

        def stacked_in_place(ts):       # the tensors already lie row-wise behind one another in one storage (this module's stack, or FusedAdam's flat buffer)
            off = 0
            for t in ts:
                if not t.is_contiguous() or t.data_ptr() != ts[0].data_ptr() + 4 * off or t.untyped_storage().data_ptr() != ts[0].untyped_storage().data_ptr():
                    return None
                off += t.numel()
            shape = (sum(t.shape[0] for t in ts),) + tuple(ts[0].shape[1:])
            return torch.empty(0, dtype=ts[0].dtype, device=ts[0].device).set_(ts[0].untyped_storage(), ts[0].storage_offset(), shape)
        with torch.no_grad():
            Wst, bst = stacked_in_place([w.data for w in ws]), stacked_in_place([b.data for b in bs])
        if Wst is not None and bst is not None:
            return Wst, bst
        with torch.no_grad():
            Wst = torch.cat([w.detach() for w in ws], dim=0).contiguous()
            bst = torch.cat([b.detach() for b in bs], dim=0).contiguous()
            off = 0
            for w, b in zip(ws, bs):
                n = w.shape[0]
                w.data = Wst[off:off + n]
                b.data = bst[off:off + n]
                off += n
        return Wst, bst

    def forward(self, feat):
        """feat: (B, d) -> list of (B, n_classes)."""
        if hasattr(self, "dropout"):
            feat = self.dropout(feat)
        # ONE GEMM for all Z future-action heads: the 20 (593, d) projections LIVE row-wise stacked in one buffer (their `.data` are views of
        # it, _stacked()), so there is no torch.cat of twenty matrices per step and no split of the gradient (round 6)
        sizes = [p.out_features for p in self.projections]
        if feat.is_cuda and len(self.projections) > 1:
            W, b = self._stacked()
            y = F_egx.StackedLinearFn.apply(feat.reshape(-1, feat.shape[-1]), W, b, sizes, self.egx_compute,
                                            *[p.weight for p in self.projections], *[p.bias for p in self.projections])
            x = list(y.view(*feat.shape[:-1], W.shape[0]).split(sizes, dim=-1))
        else:
            W = torch.cat([p.weight for p in self.projections], dim=0)
            b = torch.cat([p.bias for p in self.projections], dim=0)
            x = list(F_egx.linear(feat, W, b, self.egx_compute).split(sizes, dim=-1))
        if not self.training and not self.test_noact:
            x = [self.act(x_i) for x_i in x]
        return x


class _LTATranslator(nn.Module, TranslatorMixin):
    """What the two LTA translators share: learned positions, shared token-prep LayerNorm, post-LN encoder, token mean,
    MultiTaskHead; `generate` / `decode` / `encode_clips*` as in the reference."""

    def _build_translator(self, cfg, n_streams):
        self.cfg = cfg
        self.sequence_len = cfg.FORECASTING.NUM_INPUT_CLIPS * n_streams
        self.num_heads = cfg.MODEL.TRANSLATION_HEADS
        self.num_layers = cfg.MODEL.TRANSLATION_LAYERS
        self.feature_dim = cfg.MODEL.TRANSLATION_INPUT_FEATURES
        self.dp_rate = cfg.MODEL.TRANSLATION_DROPOUT

    def _build_encoder(self):
        self.transformer = nn.TransformerEncoder(   # parameter container only
            encoder_layer=nn.TransformerEncoderLayer(d_model=self.feature_dim, nhead=self.num_heads,
                                                     dropout=self.dp_rate, batch_first=True),
            num_layers=self.num_layers)
        self.ln = nn.LayerNorm(self.feature_dim)
        self._init_parameters()

    def _build_action_and_lta(self, cfg, lta_decoder):
        """SlowFast with a feature_dim-wide head (head stays trainable) + the frozen LTA forecasting encoder."""
        if cfg_get(cfg, "CHECKPOINT_FILE_PATH_AR"):
            self.action_model = make_hoi_backbone("slowfast", cfg=cfg, num_classes=[self.feature_dim], with_head=True,
                                                  ckpt=cfg.CHECKPOINT_FILE_PATH_AR, loader="lta")
            freeze_backbone_params(self.action_model)
        if cfg_get(cfg, "CHECKPOINT_FILE_PATH_LTA"):
            self.lta_model = make_hoi_backbone("lta", cfg=cfg, build_decoder=lta_decoder, ckpt=cfg.CHECKPOINT_FILE_PATH_LTA)
            freeze_params(self.lta_model)

    def _build_head(self, cfg):
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

    def _translate(self, feats, projs, pools=None):
        segs, off = [], 0
        for i, (f, pj) in enumerate(zip(feats, projs)):
            pool = pools[i] if pools else 1
            T = f.shape[1] // pool
            segs.append(SegmentSpec(T=T, d_in=f.shape[2], has_proj=pj is not None, add_row=None, pos_row0=off, pool=pool))
            off += T
        assert off == self.sequence_len, f"token count {off} != sequence_len {self.sequence_len}"
        tokens = self._egx_encode(feats, segs, encoder=self.transformer, ln=self.ln, projs=projs, task_embed=None,
                                  pos_table=self.pe[0], p_drop=self.dp_rate)
        pooled = F_egx.pool_head(tokens)   # mean over tokens
        return self.decode(pooled)

    def encode_clips(self, model, x):
        assert isinstance(x, list) and len(x) >= 1
        return torch.stack([model([pathway[:, i] for pathway in x]) for i in range(x[0].shape[1])], dim=1)

    def _generate(self, x, k):
        results = []
        for head_x in x:
            if k > 1:
                preds_dist = Categorical(logits=head_x)
                preds = [preds_dist.sample() for _ in range(k)]
            elif k == 1:
                preds = [head_x.argmax(2)]
            results.append(torch.stack(preds, dim=1))
        return results


@MODEL_REGISTRY.register()
class TaskFusionMFTransformerLTA4Task(_LTATranslator):
    def __init__(self, cfg):
        super().__init__()
        self._build_translator(cfg, 4)
        self.pe = nn.Parameter(torch.randn(1, self.sequence_len, self.feature_dim), requires_grad=True)
        self.proj_pnr = nn.Linear(8192, self.feature_dim)
        self.proj_oscc = nn.Linear(8192, self.feature_dim)
        self.proj_lta = nn.Linear(2048, self.feature_dim)
        self._build_encoder()
        # the four task-specific models (reference :279-302); xavier init above runs BEFORE they are attached
        if cfg_get(cfg, "PRETRAIN.PNR_CFG"):
            self.pnr_model = make_hoi_backbone("pnr", cfg_file=cfg.PRETRAIN.PNR_CFG)
            freeze_params(self.pnr_model)
        if cfg_get(cfg, "PRETRAIN.OSCC_CFG"):
            self.oscc_model = make_hoi_backbone("oscc", cfg_file=cfg.PRETRAIN.OSCC_CFG, no_temp_pool=False)
            freeze_params(self.oscc_model)
        self._build_action_and_lta(cfg, lta_decoder=True)
        self._build_head(cfg)

    def forward_features(self, feat_pnr, feat_oscc, feat_action, feat_lta):
        """pnr/oscc (B, n, 8192), action (B, n, d), lta (B, n, 2048) -> [(B, Z, #verbs), (B, Z, #nouns)]."""
        return self._translate([feat_pnr, feat_oscc, feat_action, feat_lta],
                               [self.proj_pnr, self.proj_oscc, None, self.proj_lta])

    def forward_frame_features(self, frames_pnr, frames_oscc, feat_action, feat_lta, frames_per_clip=16):
        """Feature hand-off without the pooled intermediate: frames_pnr / frames_oscc are the PNR /
        OSCC backbones' per-FRAME `middle=True` features of all clips, (B, n * frames_per_clip, 8192) in fp32 or bf16; the
        temporal mean of encode_clips_pnr (`.mean(dim=1)`) is taken on the way into the projection GEMM's bf16 operand
        (wide bf16 path). lta (B, n, 2048) as in forward_features (fp32 or bf16); action (B, n, d) must be fp32: it has no
        projection, so it enters the shared LayerNorm as it is (a packed identity segment is refused by the library)."""
        return self._translate([frames_pnr, frames_oscc, feat_action, feat_lta],
                               [self.proj_pnr, self.proj_oscc, None, self.proj_lta], pools=[frames_per_clip, frames_per_clip, 1, 1])

    def encode_clips_pnr(self, model, x):
        head = getattr(self, "_sink_heads", {}).get(id(model))
        if head is None:
            return torch.stack([model([x[:, i, ...]], middle=True).mean(dim=1) for i in range(x.shape[1])], dim=1)
        # producer side of row F4 (enable_feature_sink): the backbone's head pools, permutes, averages the frames and casts in ONE
        # pass over its res5 map, straight into token row i of the packed (B, n, 8192) stream the projection GEMM reads in place
        B, n = x.shape[0], x.shape[1]
        stream = head._stream_name
        # with autograd recording, every forward gets a stream of its own: proj_pnr / proj_oscc save it for their weight gradient,
        # and a second forward before that backward (micro-batches, an eval pass in between) would otherwise overwrite it unseen
        self._sink.alloc(stream, B, n, self.proj_pnr.in_features, fresh=torch.is_grad_enabled())
        try:
            for i in range(n):
                head.token = i
                model([x[:, i, ...]], middle=True)
        finally:
            head.token = None
        return self._sink.get(stream)

    def enable_feature_sink(self, dtype=torch.bfloat16, head_attrs=("Keyframe_localisation_head", "State_detection_head", "head")):
        """Route the PNR / OSCC backbones' `middle=True` features through a FeatureSink (egot2_amd/feature_sink.py): their pooling
        heads are swapped for PooledFeatureHead (same parameters) and `forward()` hands packed `dtype` rows to the translator.
        bf16 (default) is what the wide path's projection GEMM consumes in place; the action / LTA streams are unchanged."""
        from .feature_sink import FeatureSink, attach_sink
        self._sink = FeatureSink(self.pe.device, dtype)
        self._sink_heads = {}
        for stream, model in (("pnr", self.pnr_model), ("oscc", self.oscc_model)):
            attr = next((a for a in head_attrs if hasattr(model, a)), None)
            if attr is None:
                raise ValueError(f"enable_feature_sink: the {stream} backbone has none of the head modules {head_attrs}")
            self._sink_heads[id(model)] = attach_sink(model, attr, self._sink, stream)
        return self

    def forward(self, x_lta, x_pnr):
        # as the reference (:354-363): no no_grad() around the backbone calls; their parameters are frozen
        feat_pnr = self.encode_clips_pnr(self.pnr_model, x_pnr)
        feat_oscc = self.encode_clips_pnr(self.oscc_model, x_pnr)
        feat_action = self.encode_clips(self.action_model, x_lta)   # its head is trainable in the reference
        feat_lta = self.lta_model(x_lta, None, middle=True).transpose(0, 1)
        return self.forward_features(feat_pnr, feat_oscc, feat_action, feat_lta)

    def generate(self, x_lta, x_pnr, k=1):
        return self._generate(self.forward(x_lta, x_pnr), k)


@MODEL_REGISTRY.register()
class TaskFusionMFTransformer2Task(_LTATranslator):
    """HOI/models/lta/lta_models_lta_transfer.py:429-526: the LTA translator over the action-recognition and LTA streams
    (2 n tokens). `proj_lta` is the identity when the translator is 2048 wide, as in the reference."""

    def __init__(self, cfg):
        super().__init__()
        self._build_translator(cfg, 2)
        self.proj_lta = nn.Identity()
        if self.feature_dim != 2048:
            self.proj_lta = nn.Linear(2048, self.feature_dim)
        self.pe = nn.Parameter(torch.randn(1, self.sequence_len, self.feature_dim), requires_grad=True)
        self._build_encoder()
        self._build_action_and_lta(cfg, lta_decoder=False)
        self._build_head(cfg)

    def forward_features(self, feat_action, feat_lta):
        """action (B, n, d), lta (B, n, 2048) -> [(B, Z, #verbs), (B, Z, #nouns)]."""
        pj = self.proj_lta if isinstance(self.proj_lta, nn.Linear) else None
        return self._translate([feat_action, feat_lta], [None, pj])

    def forward(self, x, tgts=None):
        feat_action = self.encode_clips(self.action_model, x)  # (bs, num_input, d)
        feat_lta = self.lta_model(x, None, middle=True).transpose(0, 1)
        return self.forward_features(feat_action, feat_lta)

    def generate(self, x, k=1):
        return self._generate(self.forward(x), k)
