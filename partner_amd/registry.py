"""Plugin registry with the reference's semantics (det3d/utils/registry.py:6-78):
classes register under their ``__name__``; ``build_from_cfg`` pops ``type``, fills defaults
with ``setdefault`` and instantiates; unknown types raise ``KeyError``."""
from __future__ import annotations

import inspect
from typing import Any, Dict, Optional


class Registry:
    def __init__(self, name: str):
        self._name = name
        self._module_dict: Dict[str, type] = {}

    def __repr__(self):
        return f"{type(self).__name__}(name={self._name}, items={list(self._module_dict)})"

    @property
    def name(self) -> str:
        return self._name

    @property
    def module_dict(self) -> Dict[str, type]:
        return self._module_dict

    def get(self, key: str) -> Optional[type]:
        return self._module_dict.get(key)

    def register_module(self, cls):
        if not inspect.isclass(cls):
            raise TypeError(f"module must be a class, but got {type(cls)}")
        if cls.__name__ in self._module_dict:
            raise KeyError(f"{cls.__name__} is already registered in {self._name}")
        self._module_dict[cls.__name__] = cls
        return cls


def build_from_cfg(cfg: Dict[str, Any], registry: Registry, default_args: Optional[Dict[str, Any]] = None):
    if not (isinstance(cfg, dict) and "type" in cfg):
        raise AssertionError("cfg must be a dict with a 'type' key")
    if not (default_args is None or isinstance(default_args, dict)):
        raise AssertionError("default_args must be a dict or None")
    args = dict(cfg)
    kind = args.pop("type")
    if isinstance(kind, str):
        cls = registry.get(kind)
        if cls is None:
            raise KeyError(f"{kind} is not in the {registry.name} registry")
    elif inspect.isclass(kind):
        cls = kind
    else:
        raise TypeError(f"type must be a str or valid type, but got {type(kind)}")
    for k, v in (default_args or {}).items():
        args.setdefault(k, v)
    return cls(**args)
