"""Minimal stand-in for fvcore.common.registry.Registry as used by the reference
(HHI/models/ttm/build.py:7-20: MODEL_REGISTRY = Registry("MODEL"); build_model(args) looks the class up by
args.model). When fvcore is importable the real Registry is used so the classes register exactly as in the
reference tree."""
from __future__ import annotations

try:  # pragma: no cover - fvcore is not in this image
    from fvcore.common.registry import Registry  # type: ignore
except Exception:  # noqa: BLE001

    class Registry:  # type: ignore
        def __init__(self, name: str):
            self._name = name
            self._obj_map = {}

        def _do_register(self, name, obj):
            assert name not in self._obj_map, f"An object named '{name}' was already registered in '{self._name}' registry!"
            self._obj_map[name] = obj

        def register(self, obj=None):
            if obj is None:
                def deco(func_or_class):
                    self._do_register(func_or_class.__name__, func_or_class)
                    return func_or_class
                return deco
            self._do_register(obj.__name__, obj)

        def get(self, name):
            ret = self._obj_map.get(name)
            if ret is None:
                raise KeyError(f"No object named '{name}' found in '{self._name}' registry!")
            return ret

        def __contains__(self, name):
            return name in self._obj_map


def make_registry(name: str = "MODEL") -> "Registry":
    reg = Registry(name)
    reg.__doc__ = "Registry for video modeling (drop-in for the reference's MODEL_REGISTRY)."
    return reg


def build_model_from(registry, args_or_cfg, name: str):
    """Reference semantics: MODEL_REGISTRY.get(name)(args)."""
    return registry.get(name)(args_or_cfg)
