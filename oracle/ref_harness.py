"""ORACLE — test infrastructure only. Imports the REAL reference translator classes from /root/reference with
import-time stubs for packages this image lacks (fvcore Registry, torchtext.vocab, torchaudio). No arithmetic is
stubbed: the reference's own forward() runs, fed by pass-through "backbones" that hand the synthetic feature
tensors straight to the translator.

Only usable where /root/reference exists (this container); the GPU box never sees it. HHI and HOI share
top-level package names (`models`, `utils`), so use one tree per process.
"""
from __future__ import annotations

import os
import sys
import types
from argparse import Namespace

import torch
import torch.nn as nn

REFERENCE_ROOT = os.environ.get("EGOT2_REFERENCE", "/root/reference")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "HHI", "models"))


def _stub_module(name: str, **attrs):
    m = sys.modules.get(name)
    if m is None:
        m = types.ModuleType(name)
        sys.modules[name] = m
    for k, v in attrs.items():
        setattr(m, k, v)
    return m


def _install_stubs():
    class Registry:  # the 10-line dict registry of fvcore
        def __init__(self, name):
            self._name, self._obj_map = name, {}

        def register(self, obj=None):
            if obj is None:
                def deco(o):
                    self._obj_map[o.__name__] = o
                    return o
                return deco
            self._obj_map[obj.__name__] = obj

        def get(self, name):
            return self._obj_map[name]

    try:
        import fvcore  # noqa: F401
    except ImportError:
        _stub_module("fvcore")
        _stub_module("fvcore.common")
        _stub_module("fvcore.common.registry", Registry=Registry)
    try:
        import torchtext  # noqa: F401
    except ImportError:
        _stub_module("torchtext")
        _stub_module("torchtext.vocab", vocab=lambda *a, **k: None, build_vocab_from_iterator=lambda *a, **k: None)
    try:
        import torchaudio  # noqa: F401
    except ImportError:
        _stub_module("torchaudio")
        _stub_module("torchaudio.transforms", MelSpectrogram=lambda *a, **k: nn.Identity())
        sys.modules["torchaudio"].transforms = sys.modules["torchaudio.transforms"]


def use_tree(tree: str):
    """Put /root/reference/<tree> (HHI or HOI) first on sys.path."""
    if not reference_available():
        raise RuntimeError(f"reference tree not found under {REFERENCE_ROOT}")
    _install_stubs()
    path = os.path.join(REFERENCE_ROOT, tree)
    other = "HOI" if tree == "HHI" else "HHI"
    if any(p.rstrip("/").endswith("/" + other) for p in sys.path):
        raise RuntimeError("HHI and HOI reference trees cannot be mixed in one process")
    if path not in sys.path:
        sys.path.insert(0, path)


# ---- pass-through backbones (feature tensors travel through the reference's own forward) ----------
class _LamPass(nn.Module):
    def forward(self, video, middle=False):
        return video


class _TtmPass(nn.Module):
    def forward(self, video, audio, middle=False):
        return audio


class _AsdPass(nn.Module):
    def forward_audio_frontend(self, x):
        return x

    def forward_visual_frontend(self, x):
        return x

    def forward_cross_attention(self, a, v):
        return a, v

    def forward_audio_visual_backend(self, a, v):
        return a.reshape(-1, a.shape[-1])


def hhi_args(hidden_dim=128, num_heads=4, dropout=0.0, num_layers=1, **kw) -> Namespace:
    return Namespace(lam_checkpoint=None, ttm_checkpoint=None, asd_checkpoint=None, nofreeze=True,
                     hidden_dim=hidden_dim, num_heads=num_heads, dropout=dropout, num_layers=num_layers,
                     hidden_dim2=512, **kw)


def attach_passthrough(model: nn.Module) -> nn.Module:
    model.lam_model, model.ttm_model, model.asd_model = _LamPass(), _TtmPass(), _AsdPass()
    return model


def ref_ttm(n_tasks: int, args: Namespace) -> nn.Module:
    """Real reference TaskFusionMFTransformer{2,3}Task (HHI/models/ttm/model_taskspecific.py:154,197)."""
    use_tree("HHI")
    from models.ttm.model_taskspecific import TaskFusionMFTransformer2Task, TaskFusionMFTransformer3Task
    cls = TaskFusionMFTransformer2Task if n_tasks == 2 else TaskFusionMFTransformer3Task
    return attach_passthrough(cls(args))


def ref_ttm_forward(model: nn.Module, ttm_out, lam_out, asd_out=None):
    """Calls the reference forward(): lam_model(video)->video, ttm_model(video, audio)->audio,
    asd backend(audio_asd)->audio_asd; video_asd only provides (N, D, H, W) = (B, T, 1, 1)."""
    if asd_out is None:
        return model(lam_out, ttm_out)  # forward(video, audio)
    B, T = asd_out.shape[:2]
    return model(lam_out, torch.zeros(B, T, 1, 1), ttm_out, asd_out)


def ref_asd(args: Namespace) -> nn.Module:
    """Real reference HHI/models/asd/model_taskspecific.py:108 TaskFusionMFTransformer3Task."""
    use_tree("HHI")
    from models.asd.model_taskspecific import TaskFusionMFTransformer3Task
    return attach_passthrough(TaskFusionMFTransformer3Task(args))


def ref_hhi_g(args: Namespace, vocab=None) -> nn.Module:
    """Real reference TaskTranslationPromptTransformer (HHI/models/multitask/task_prompt_model.py:174) with the
    backbone constructors patched out and CustomDecoderLayer._mha_block adapted to torch>=2 (extra is_causal arg;
    arithmetic unchanged)."""
    use_tree("HHI")
    import models.multitask.task_prompt_model as tpm
    tpm.LAMBackbone = lambda ckpt: _LamPass()
    tpm.TTMBackbone = lambda ckpt: _TtmPass()
    tpm.talkNetModel = lambda: _AsdPass()
    tpm.load_ckpt = lambda *a, **k: None
    tpm.freeze_params = lambda m: None
    orig = tpm.CustomDecoderLayer._mha_block

    def _mha_block(self, x, mem, attn_mask, key_padding_mask, is_causal=False):
        return orig(self, x, mem, attn_mask, key_padding_mask)

    if getattr(orig, "__name__", "") == "_mha_block" and orig.__code__.co_argcount == 5:
        tpm.CustomDecoderLayer._mha_block = _mha_block
    vocab = vocab or {'</s>': 0, '<unk>': 1, 'ttm': 2, 'lam': 3, 'asd': 4, '0': 5, '1': 6}
    return tpm.TaskTranslationPromptTransformer(args, vocab)


# ---- HOI tree ------------------------------------------------------------------------------------------
def _install_hoi_stubs():
    _install_stubs()

    class CfgNode(dict):
        pass

    try:
        import fvcore.nn  # noqa: F401
    except ImportError:
        _stub_module("fvcore.nn")
        _stub_module("fvcore.nn.weight_init", c2_msra_fill=lambda m: None, c2_xavier_fill=lambda m: None)
        sys.modules["fvcore.nn"].weight_init = sys.modules["fvcore.nn.weight_init"]
        _stub_module("fvcore.common.config", CfgNode=CfgNode)
    try:
        import detectron2  # noqa: F401
    except ImportError:
        _stub_module("detectron2")
        _stub_module("detectron2.layers", ROIAlign=lambda *a, **k: nn.Identity())


class _FeatPass(nn.Module):
    """Stands in for a frozen HOI backbone: returns the tensor it is called with, whatever the call protocol."""

    def forward(self, x, *a, **k):
        return x[0] if isinstance(x, (list, tuple)) else x


def hoi_cfg(d=256, heads=8, layers=2, n_clips=4, num_classes=(5, 7), z=3, dropout=0.0):
    from types import SimpleNamespace as NS
    return NS(FORECASTING=NS(NUM_INPUT_CLIPS=n_clips, NUM_ACTIONS_TO_PREDICT=z),
              MODEL=NS(TRANSLATION_HEADS=heads, TRANSLATION_LAYERS=layers, TRANSLATION_INPUT_FEATURES=d,
                       TRANSLATION_DROPOUT=dropout, NUM_CLASSES=list(num_classes), DROPOUT_RATE=0.0, HEAD_ACT="softmax"),
              TEST=NS(NO_ACT=False), PRETRAIN=NS(PNR_CFG=None, OSCC_CFG=None),
              CHECKPOINT_FILE_PATH_AR=None, CHECKPOINT_FILE_PATH_LTA=None)


class _SecondT(nn.Module):
    """ForecastingEncoderDecoder stand-in for the LTA 4-task model: `lta_model(x_lta, None, middle=True)` returns the
    SECOND pathway tensor (B, n, 2048) as (n, B, 2048) (indexing only) — the reference transposes it back."""

    def forward(self, x, *a, **k):
        return x[1].transpose(0, 1)


def ref_lta4(cfg) -> nn.Module:
    """Real reference TaskFusionMFTransformerLTA4Task (HOI/models/lta/lta_models_lta_transfer.py:257): constructors of
    the four frozen backbones and their checkpoint loaders are patched out; __init__ AND forward(x_lta, x_pnr)
    (:354-363, incl. encode_clips / encode_clips_pnr with its `.mean(dim=1)` over frames) are the reference's.
    Call it as `model([action (B, n, d), lta (B, n, 2048)], frames (B, n, F, 8192))`: the PNR stand-in hands back the
    clip's frames, the OSCC stand-in the same frames reversed along the channel axis (the reference feeds both
    backbones the same x_pnr), the SlowFast stand-in the first pathway's clip, the LTA stand-in the second pathway."""
    use_tree("HOI")
    _install_hoi_stubs()
    import models.lta.lta_models_lta_transfer as m
    from types import SimpleNamespace as NS
    m.load_pnr_config = lambda path: NS(MISC=NS(CHECKPOINT_FILE_PATH=None), MODEL=NS(NO_TEMP_POOL=False))
    m.KeyframeLocalizationResNet = lambda cfg: _FeatPass()
    m.StateChangeClsResNet = lambda cfg: _Flip()
    m.SlowFast = lambda cfg, with_head=True: _FeatPass()
    m.ForecastingEncoderDecoder = lambda cfg, build_decoder=True: _SecondT()
    m.load_ckpt = lambda *a, **k: None
    m.load_lta_backbone = lambda *a, **k: None
    m.freeze_params = lambda *a, **k: None
    m.freeze_backbone_params = lambda *a, **k: None
    return m.TaskFusionMFTransformerLTA4Task(cfg)


# ---- HOI EgoT2-s: PNR/OSCC translator and the two action-recognition translators -------------
class _Pick(nn.Module):
    """Frozen-backbone stand-in that returns element `i` of the list it is called with (the reference hands the same
    input list to the PNR and OSCC backbones: `x_oscc = x_pnr.copy()`)."""

    def __init__(self, i):
        super().__init__()
        self.i = i

    def forward(self, x, *a, **k):
        return x[self.i]


class _ListPass(nn.Module):
    """SlowFast stand-in: `model(x, middle=True)` returns the [slow, fast] pathway list unchanged."""

    def forward(self, x, *a, **k):
        return list(x)


class _LtaPass(nn.Module):
    """ForecastingEncoderDecoder stand-in: (n, B, 2048) clip features = the slow input's first spatial element
    (indexing only, no arithmetic), transposed the way the reference expects it."""

    def forward(self, x, *a, **k):
        return x[0][:, :, :, 0, 0, 0].transpose(0, 1)


def hoi_s_cfg(d=128, layers=2, heads=8, num_classes=(5, 7), task="state_change_detection", n_clips=2, dropout=0.0,
              feat_dropout=0.0):
    from types import SimpleNamespace as NS
    return NS(DATA=NS(TASK=task),
              FORECASTING=NS(NUM_INPUT_CLIPS=n_clips, INPUT_OFFSET=0),
              MODEL=NS(TRANSLATION_INPUT_FEATURES=d, TRANSLATION_LAYERS=layers, TRANSLATION_HEADS=heads,
                       TRANSLATION_DROPOUT=dropout, FEAT_DROPOUT_RATE=feat_dropout, TRANSFORMER_DROPOUT_RATE=dropout,
                       NUM_CLASSES=list(num_classes)),
              PRETRAIN=NS(PNR_CFG=None, OSCC_CFG=None, ACTION_CFG=None, LTA_CFG=None, PNR_FT=True, OSCC_FT=True, ACTION_FT=True))


def ref_pnr3(cfg) -> nn.Module:
    """Real TaskFusionMFTransformer3TaskDropout (HOI/models/pnr/video_model_transfer_3task.py:212-258). With the
    PRETRAIN.*_CFG entries None the base class builds no backbone; pass-through stand-ins are attached instead and the
    REAL forward(x1, x2) runs on x1 = [pnr_feat, oscc_feat], x2 = [slow (B,2048,8,1,1), fast (B,256,8,1,1)]."""
    use_tree("HOI")
    _install_hoi_stubs()
    import models.pnr.video_model_transfer_3task as m
    model = m.TaskFusionMFTransformer3TaskDropout(cfg)
    model.pnr_model, model.oscc_model, model.recognition_model = _Pick(0), _Pick(1), _ListPass()
    return model


def ref_ar3(cfg) -> nn.Module:
    """Real TaskFusionMFTransformer3Task of the action-recognition task (HOI/models/lta/lta_models_transfer.py:96-137):
    forward(x_action=[slow, fast], x_pnr=[pnr_feat, oscc_feat]) -> [verb logits, noun logits]."""
    use_tree("HOI")
    _install_hoi_stubs()
    import models.lta.lta_models_transfer as m
    model = m.TaskFusionMFTransformer3Task(cfg)
    model.pnr_model, model.oscc_model, model.recognition_model = _Pick(0), _Pick(1), _ListPass()
    return model


def ref_ar2(cfg) -> nn.Module:
    """Real TaskFusionMFTransformer2TaskAR (HOI/models/lta/lta_models_transfer.py:170-235); the constructors and
    checkpoint loaders of its two frozen backbones are patched out. forward(x) with x = [slow (B, n+1, 2048, 8, 1, 1),
    fast (B, n+1, 256, 8, 1, 1)]: the last clip feeds the action pathway, the first n clips the LTA pathway."""
    use_tree("HOI")
    _install_hoi_stubs()
    import models.lta.lta_models_transfer as m
    from types import SimpleNamespace as NS
    m.load_lta_config = lambda path: NS(MODEL=NS(NUM_CLASSES=None, HEAD_ACT=None), CHECKPOINT_FILE_PATH=None)
    m.SlowFast = lambda cfg, with_head=True: _ListPass()
    m.ForecastingEncoderDecoder = lambda cfg, build_decoder=True: _LtaPass()
    m.load_lta_backbone = lambda *a, **k: None
    m.freeze_params = lambda *a, **k: None
    m.freeze_backbone_params = lambda *a, **k: None
    return m.TaskFusionMFTransformer2TaskAR(cfg)


def pathway5d(feat: torch.Tensor) -> torch.Tensor:
    """(B, T, C) pooled pathway features -> the (B, C, T, 1, 1) tensor a SlowFast pathway would hand over (the
    reference's adaptive average pools are the identity on it when T already has the pooled length)."""
    return feat.permute(0, 2, 1)[..., None, None].contiguous()


# ---- HOI EgoT2-g (config C5) ----------------------------------------------------------------------------------------
class _Flip(nn.Module):
    """OSCC stand-in for the 'lta' prompts, where the reference feeds PNR and OSCC the same clip tensor
    (`copy.deepcopy(video_pnr)`): returns its input reversed along the channel axis (indexing only) so that the two
    feature streams differ."""

    def forward(self, x, *a, **k):
        return x[0].flip(-1)


class _AcPass(nn.Module):
    """Recognition-model stand-in with the two call protocols of the 6-task model: `model(x, middle=True)` -> the
    [slow, fast] pathway list; `model([a_i, l_i])` (per-clip, with head) -> a_i."""

    def forward(self, x, *a, middle=False, **k):
        return list(x) if middle else x[0]


class _LtaPass2(nn.Module):
    """`lta_model(video_ac, None, middle=True)` -> (n, B, 2048): the second element of video_ac, transposed."""

    def forward(self, x, *a, **k):
        return x[1].transpose(0, 1)


def hoi_g_args(hidden_dim=256, num_heads=8, num_layers=2, dropout=0.0):
    return Namespace(hidden_dim=hidden_dim, num_heads=num_heads, num_layers=num_layers, dropout=dropout,
                     pnr_cfg_file=None, oscc_cfg_file=None, action_cfg_file=None, lta_cfg_file=None)


HOI_G_VOCAB = {'</s>': 0, '<unk>': 1, 'pnr': 2, 'oscc': 3, 'action_verb': 4, 'action_noun': 5, 'lta_verb': 6, 'lta_noun': 7,
               '0': 8, '1': 9, '2': 10, '3': 11}


def ref_hoi_g(args, vocab=None) -> nn.Module:
    """Real TaskTranslationPromptTransformer6Task (HOI/models/multitask/video_model_builder.py:278-383); config loaders,
    backbone constructors and checkpoint loaders are patched out, the encoder arithmetic is the reference's. Call
    `model.encode(video_pnr, video_ac, task)`:
      other tasks: video_pnr = [pnr_feat (B,16,8192), oscc_feat], video_ac = [slow (B,2048,8,1,1), fast (B,256,8,1,1)]
      'lta*'     : video_pnr = (B, n, 1, 8192) tensor, video_ac = [action (B, n, d), lta (B, n, 2048)]."""
    use_tree("HOI")
    _install_hoi_stubs()
    import models.multitask.video_model_builder as m
    from types import SimpleNamespace as NS
    mk = lambda: NS(MISC=NS(CHECKPOINT_FILE_PATH=None), MODEL=NS(NO_TEMP_POOL=False, NUM_CLASSES=None, HEAD_ACT=None),  # noqa: E731
                    CHECKPOINT_FILE_PATH=None, CHECKPOINT_FILE_PATH_LTA=None, FORECASTING=NS(NUM_ACTIONS_TO_PREDICT=20))
    m.load_config_file = lambda path: mk()
    m.load_lta_config = lambda path: mk()
    m.KeyframeLocalizationResNet = lambda cfg: _Pick(0)
    m.StateChangeClsResNet = lambda cfg: _Pick(1)
    m.SlowFast = lambda cfg, with_head=True: _AcPass()
    m.ForecastingEncoderDecoder = lambda cfg, build_decoder=True: _LtaPass2()
    for name in ("load_checkpoint", "load_lta_backbone", "freeze_params", "load_recognition_backbone", "freeze_backbone_params"):
        setattr(m, name, lambda *a, **k: None)
    orig = m.CustomDecoderLayer._mha_block
    if orig.__code__.co_argcount == 5:      # torch >= 2 passes is_causal

        def _mha_block(self, x, mem, attn_mask, key_padding_mask, is_causal=False):
            return orig(self, x, mem, attn_mask, key_padding_mask)
        m.CustomDecoderLayer._mha_block = _mha_block
    return m.TaskTranslationPromptTransformer6Task(args, vocab or HOI_G_VOCAB)


def hoi_g_encode_lta(model, feat_pnr_clips, feat_action, feat_lta):
    """Runs the REAL encode() on the 'lta' branch. feat_pnr_clips (B, n, 1, 8192): the reference averages each clip's
    frames itself (`.mean(dim=1)`, one frame here); the OSCC stream is the channel-reversed copy (see _Flip)."""
    model.pnr_model, model.oscc_model = _Pick(0), _Flip()
    return model.encode(feat_pnr_clips, [feat_action, feat_lta], "lta_verb")


def hoi_g_encode_other(model, task, feat_pnr, feat_oscc, slow, fast):
    model.pnr_model, model.oscc_model = _Pick(0), _Pick(1)
    return model.encode([feat_pnr, feat_oscc], [pathway5d(slow), pathway5d(fast)], task)


# ---- F3 remainder: LTA 2-task translator, the pre-LN PNR translator, the 2-task and action EgoT2-g models ---------------
def ref_lta2(cfg) -> nn.Module:
    """Real TaskFusionMFTransformer2Task of the LTA task (HOI/models/lta/lta_models_lta_transfer.py:429-526); backbone
    constructors / loaders patched out. REAL forward(x) with x = [action (B, n, d), lta (B, n, 2048)]: the per-clip
    action model returns its first input, the LTA model the second one (transposed the way the reference expects)."""
    use_tree("HOI")
    _install_hoi_stubs()
    import models.lta.lta_models_lta_transfer as m
    m.SlowFast = lambda cfg, with_head=True: _AcPass()
    m.ForecastingEncoderDecoder = lambda cfg, build_decoder=True: _LtaPass2()
    for name in ("load_ckpt", "load_lta_backbone", "freeze_params", "freeze_backbone_params"):
        setattr(m, name, lambda *a, **k: None)
    return m.TaskFusionMFTransformer2Task(cfg)


def ref_pnrvit(task="keyframe_localization") -> nn.Module:
    """Real TaskFusionMFTransformer (HOI/models/pnr/video_model_transfer.py:44-67) over the real simple_vit.Transformer; with
    empty PRETRAIN.*_CFG the base class builds no backbone. REAL forward(x) on x = [pnr_feat, oscc_feat]."""
    use_tree("HOI")
    _install_hoi_stubs()
    import models.pnr.video_model_transfer as m
    from types import SimpleNamespace as NS
    model = m.TaskFusionMFTransformer(NS(DATA=NS(TASK=task), PRETRAIN=NS(PNR_CFG=None, OSCC_CFG=None, PNR_FT=True, OSCC_FT=True)))
    model.pnr_model, model.oscc_model = _Pick(0), _Pick(1)
    return model


def _patch_decoder_layer(m):
    orig = m.CustomDecoderLayer._mha_block
    if orig.__code__.co_argcount == 5:      # torch >= 2 passes is_causal

        def _mha_block(self, x, mem, attn_mask, key_padding_mask, is_causal=False):
            return orig(self, x, mem, attn_mask, key_padding_mask)
        m.CustomDecoderLayer._mha_block = _mha_block


def ref_hoi_g2(args, vocab=None) -> nn.Module:
    """Real TaskTranslationPromptTransformer2Task (HOI/models/multitask/video_model_builder_2task.py:124-167). REAL
    encode(video_pnr) on video_pnr = [pnr_feat, oscc_feat]."""
    use_tree("HOI")
    _install_hoi_stubs()
    import models.multitask.video_model_builder_2task as m
    from types import SimpleNamespace as NS
    m.load_config_file = lambda path: NS(MISC=NS(CHECKPOINT_FILE_PATH=None), MODEL=NS(NO_TEMP_POOL=False))
    m.KeyframeLocalizationResNet = lambda cfg: _Pick(0)
    m.StateChangeClsResNet = lambda cfg: _Pick(1)
    for name in ("load_checkpoint", "freeze_params"):
        setattr(m, name, lambda *a, **k: None)
    _patch_decoder_layer(m)
    return m.TaskTranslationPromptTransformer2Task(args, vocab or HOI_G_VOCAB)


class _AcClip(nn.Module):
    """action model of the ActionTask EgoT2-g: `model(video)` on the non-lta prompts gets the [action, lta] list and returns
    clip 0 of the action stream (B, d); per clip (`model([a_i, l_i])`, 2-D inputs) it returns a_i."""

    def forward(self, x, *a, **k):
        return x[0][:, 0] if x[0].dim() == 3 else x[0]


def ref_hoi_ga(args, vocab=None) -> nn.Module:
    """Real TaskTranslationPromptTransformerActionTask (HOI/models/multitask/video_model_builder_action.py:21-187). REAL
    encode(video, task) on video = [action (B, n, d), lta (B, n, d)]."""
    use_tree("HOI")
    _install_hoi_stubs()
    import models.multitask.video_model_builder_action as m
    from types import SimpleNamespace as NS
    m.load_lta_config = lambda path: NS(MODEL=NS(NUM_CLASSES=None, HEAD_ACT=None), CHECKPOINT_FILE_PATH_AR=None,
                                        CHECKPOINT_FILE_PATH_LTA=None, FORECASTING=NS(NUM_SEQUENCES_TO_PREDICT=1, NUM_ACTIONS_TO_PREDICT=20))
    m.vocab_idx_to_orig = lambda: (None, None)
    m.SlowFast = lambda cfg, with_head=True: _AcClip()
    m.ForecastingEncoderDecoder = lambda cfg, build_decoder=True: _LtaPass2()
    for name in ("load_checkpoint", "load_lta_backbone", "freeze_params", "freeze_backbone_params"):
        setattr(m, name, lambda *a, **k: None)
    _patch_decoder_layer(m)
    args.ff_dim = 2048
    return m.TaskTranslationPromptTransformerActionTask(args, vocab or HOI_G_VOCAB)


# ---- HOI: the PNR / OSCC backbones' head, producer side of the feature hand-off -------------------
def ref_pnr_head(num_classes: int, pool_size, act_func: str = "softmax_2", dropout_rate: float = 0.0) -> nn.Module:
    """The REAL `ResNetKeyframeLocalizationHead` (HOI/models/pnr/head_helper.py:293-381; built at
    HOI/models/pnr/video_model_builder.py:310-322 with pool (1, 7, 7) for keyframe localisation and :350-362 with
    (T, 7, 7) for state-change classification). Only the import of detectron2's ROIAlign (used by a different head of the
    same file) is stubbed; `forward(inputs, middle)` is the reference's own."""
    use_tree("HOI")
    _install_hoi_stubs()
    from models.pnr import head_helper
    return head_helper.ResNetKeyframeLocalizationHead(dim_in=[2048], num_classes=num_classes, pool_size=[list(pool_size)],
                                                      dropout_rate=dropout_rate, act_func=act_func)
