"""Frozen Stage-I backbones are OUT of this package's scope: they stay the reference's own
stock-PyTorch modules. This file only holds the glue the translator constructors need: the freeze helper and a
factory that resolves the reference's backbone classes when the translator is dropped into the reference tree
(HHI/ or HOI/ on sys.path), or accepts injected modules.
"""
from __future__ import annotations

import importlib
from typing import Callable, Dict

import torch.nn as nn


def freeze_params(model: nn.Module):
    """Same behaviour as HHI/utils/utils.py freeze_params: no grads + eval mode left to the caller."""
    for p in model.parameters():
        p.requires_grad = False


_FACTORIES: Dict[str, Callable] = {}


def register_backbone_factory(kind: str, fn: Callable):
    """kind in {'lam', 'ttm', 'asd'}; fn(ckpt_path) -> nn.Module with the reference call protocol."""
    _FACTORIES[kind] = fn


def make_backbone(kind: str, ckpt):
    if kind in _FACTORIES:
        return _FACTORIES[kind](ckpt)
    try:
        if kind == "lam":
            return importlib.import_module("models.lam.model").LAMBackbone(ckpt)
        if kind == "ttm":
            return importlib.import_module("models.ttm.model").TTMBackbone(ckpt)
        if kind == "asd":
            m = importlib.import_module("models.asd.talkNetModel").talkNetModel()
            importlib.import_module("utils.utils").load_ckpt(m, ckpt, load_asd=True)
            return m
    except ImportError as e:
        raise ImportError(
            f"a '{kind}' checkpoint was given but the reference backbone classes are not importable ({e}). "
            "Run inside the reference HHI/ tree, register a factory with "
            "egot2_amd.backbones.register_backbone_factory, or pass *_checkpoint=None and attach modules "
            "(model.lam_model = ...) / call forward_features() with precomputed features.") from e
    raise KeyError(kind)


# ---- HOI tree ---------------------------------------------------------------------------------------------------------
# The HOI translator constructors build their frozen backbones from yacs config files and checkpoints
# (HOI/models/lta/lta_models_lta_transfer.py:279-302, HOI/models/pnr/video_model_transfer_3task.py:23-58,
# HOI/models/multitask/video_model_builder.py:98-130). `make_hoi_backbone(kind, ...)` does the same through the reference's own
# classes and loaders when the HOI tree is importable, or through a factory registered with
# register_backbone_factory("hoi_<kind>", fn): fn receives the keyword arguments listed below and returns the module.
#
#   kind        keyword arguments                                   reference construction
#   pnr         cfg_file                                            KeyframeLocalizationResNet(load_config_file(cfg_file)) + load_checkpoint
#   oscc        cfg_file, no_temp_pool                              StateChangeClsResNet(cfg with MODEL.NO_TEMP_POOL) + load_checkpoint
#   slowfast    cfg | cfg_file, num_classes, with_head, ckpt, loader   SlowFast(cfg', with_head) + load_lta_backbone / load_recognition_backbone
#   lta         cfg | cfg_file, build_decoder, ckpt                 ForecastingEncoderDecoder(cfg, build_decoder) + load_lta_backbone
def freeze_backbone_params(model: nn.Module):
    """HOI/utils/multitask/load_model.py freeze_backbone_params: everything but the head stays frozen."""
    for name, p in model.named_parameters():
        if "head" not in name:
            p.requires_grad = False


def make_hoi_backbone(kind: str, **kw):
    key = "hoi_" + kind
    if key in _FACTORIES:
        return _FACTORIES[key](**kw)
    try:
        import copy
        lm = importlib.import_module("utils.multitask.load_model")
        if kind in ("pnr", "oscc"):
            cfg = importlib.import_module("utils.pnr.parser").load_config_file(kw["cfg_file"])
            vb = importlib.import_module("models.pnr.video_model_builder")
            if kind == "oscc":
                cfg.MODEL.NO_TEMP_POOL = bool(kw.get("no_temp_pool", False))
            model = (vb.KeyframeLocalizationResNet if kind == "pnr" else vb.StateChangeClsResNet)(cfg)
            lm.load_checkpoint(model, cfg.MISC.CHECKPOINT_FILE_PATH)
            return model
        if kind == "slowfast":
            cfg = kw.get("cfg")
            if cfg is None:
                cfg = importlib.import_module("utils.lta.parser").load_config_from_file(kw["cfg_file"])
            cfg = copy.deepcopy(cfg)
            cfg.MODEL.NUM_CLASSES = list(kw["num_classes"])
            cfg.MODEL.HEAD_ACT = None
            model = importlib.import_module("models.lta.video_model_builder").SlowFast(cfg, with_head=kw.get("with_head", True))
            ckpt = kw.get("ckpt", getattr(cfg, "CHECKPOINT_FILE_PATH", None))
            if kw.get("loader", "lta") == "recognition":
                lm.load_recognition_backbone(model, ckpt)
            else:
                lm.load_lta_backbone(model, ckpt, True, True)
            return model
        if kind == "lta":
            cfg = kw.get("cfg")
            if cfg is None:
                cfg = importlib.import_module("utils.lta.parser").load_config_from_file(kw["cfg_file"])
            model = importlib.import_module("models.lta.lta_models").ForecastingEncoderDecoder(cfg, build_decoder=kw.get("build_decoder", True))
            lm.load_lta_backbone(model, kw.get("ckpt", getattr(cfg, "CHECKPOINT_FILE_PATH_LTA", None)))
            return model
    except ImportError as e:
        raise ImportError(
            f"the HOI '{kind}' backbone was requested but the reference classes are not importable ({e}). Run inside the "
            f"reference HOI/ tree, register a factory with egot2_amd.backbones.register_backbone_factory('hoi_{kind}', fn), or "
            "leave the config entry empty and attach the module / call forward_features() with precomputed features.") from e
    raise KeyError(kind)


def cfg_get(cfg, path: str, default=None):
    """cfg.A.B.C with a default when any level is missing (yacs nodes and SimpleNamespace alike)."""
    cur = cfg
    for part in path.split("."):
        if cur is None or not hasattr(cur, part):
            return default
        cur = getattr(cur, part)
    return cur
