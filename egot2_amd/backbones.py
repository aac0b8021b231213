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
