"""Config files: a python file is imported and its public module-level names become an
attribute-accessible dict (behaviour of det3d/torchie/utils/config.py:12-162, written without
the third-party ``addict`` package)."""
from __future__ import annotations

import os
import sys
from importlib import import_module


class ConfigDict(dict):
    """dict with attribute access; nested dicts are converted; missing keys raise
    KeyError on item access and AttributeError on attribute access."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        for k, v in dict(*args, **kwargs).items():
            self[k] = v

    @classmethod
    def _wrap(cls, v):
        if isinstance(v, dict) and not isinstance(v, ConfigDict):
            return cls(v)
        if isinstance(v, (list, tuple)):
            return type(v)(cls._wrap(e) for e in v)
        return v

    def __setitem__(self, k, v):
        super().__setitem__(k, self._wrap(v))

    def __setattr__(self, k, v):
        self[k] = v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(f"'{type(self).__name__}' object has no attribute '{k}'") from None

    def __delattr__(self, k):
        try:
            del self[k]
        except KeyError:
            raise AttributeError(k) from None


class Config:
    @staticmethod
    def fromfile(filename: str) -> "Config":
        filename = os.path.abspath(os.path.expanduser(filename))
        if not os.path.isfile(filename):
            raise FileNotFoundError(f'file "{filename}" does not exist')
        if not filename.endswith(".py"):
            raise IOError("Only py type is supported")
        name = os.path.basename(filename)[:-3]
        if "." in name:
            raise ValueError("Dots are not allowed in config file path.")
        sys.path.insert(0, os.path.dirname(filename))
        try:
            sys.modules.pop(name, None)
            mod = import_module(name)
        finally:
            sys.path.pop(0)
        cfg = {k: v for k, v in mod.__dict__.items() if not k.startswith("__")}
        return Config(cfg, filename=filename)

    def __init__(self, cfg_dict=None, filename=None):
        cfg_dict = {} if cfg_dict is None else cfg_dict
        if not isinstance(cfg_dict, dict):
            raise TypeError(f"cfg_dict must be a dict, but got {type(cfg_dict)}")
        object.__setattr__(self, "_cfg_dict", ConfigDict(cfg_dict))
        object.__setattr__(self, "_filename", filename)
        text = ""
        if filename:
            with open(filename, "r") as f:
                text = f.read()
        object.__setattr__(self, "_text", text)

    filename = property(lambda self: self._filename)
    text = property(lambda self: self._text)

    def __repr__(self):
        return f"Config (path: {self._filename}): {self._cfg_dict!r}"

    def __len__(self):
        return len(self._cfg_dict)

    def __getattr__(self, name):
        return getattr(self._cfg_dict, name)

    def __getitem__(self, name):
        return self._cfg_dict[name]

    def __setattr__(self, name, value):
        self._cfg_dict[name] = value

    def __setitem__(self, name, value):
        self._cfg_dict[name] = value

    def __iter__(self):
        return iter(self._cfg_dict)

    def __contains__(self, name):
        return name in self._cfg_dict


def get_downsample_factor(model_config) -> int:
    """prod(ds_layer_strides) / us_layer_strides[-1] * backbone.ds_factor
    (det3d/utils/config_tool.py:39-53)."""
    import numpy as np

    try:
        neck = model_config["neck"]
    except Exception:
        model_config = model_config["first_stage_cfg"]
        neck = model_config["neck"]
    f = np.prod(neck.get("ds_layer_strides", [1]))
    if len(neck.get("us_layer_strides", [])) > 0:
        f = f / neck.get("us_layer_strides", [])[-1]
    f = int(f * model_config["backbone"]["ds_factor"])
    assert f > 0
    return f
