"""Config -> module factory with the reference's calling convention.

`Cls.init(config, **overrides)` (tts/modules/constructor.py:41-84 of the reference): merge a config (dict-like,
dataclass or None) with keyword overrides, drop service keys (leading "_"), drop keys the constructor does not take
(with a warning), refuse MISSING ("???") values, then call the constructor.  OmegaConf is not a dependency here:
any Mapping (OmegaConf's DictConfig included) works.
"""
from __future__ import annotations

import dataclasses
import warnings
from collections.abc import Mapping
from inspect import signature

MISSING = "???"


@dataclasses.dataclass
class ModuleConfig:
    def get(self, key, default=None):
        return getattr(self, key, default)

    def to_dict(self) -> dict:
        return {f.name: getattr(self, f.name) for f in dataclasses.fields(self)}


def _plain(cfg) -> dict:
    if cfg is None:
        return {}
    if dataclasses.is_dataclass(cfg) and not isinstance(cfg, type):
        return {f.name: getattr(cfg, f.name) for f in dataclasses.fields(cfg)}
    if isinstance(cfg, Mapping):
        return dict(cfg)
    raise TypeError(f"unsupported config type {type(cfg)}")


class Constructor:
    @classmethod
    def init(cls, config=None, **parameters):
        merged = _plain(config)
        merged.update(parameters)
        merged = {k: v for k, v in merged.items() if not str(k).startswith("_")}
        sig = dict(signature(cls.__init__).parameters)
        if "kwargs" not in sig:
            unknown = [k for k in merged if k not in sig]
            if unknown:
                warnings.warn(f"{cls.__name__}: ignoring parameters the constructor does not take: {unknown}")
                merged = {k: v for k, v in merged.items() if k not in unknown}
        missing = [k for k, v in merged.items() if isinstance(v, str) and v == MISSING]
        if missing:
            raise RuntimeError(f"The following params are mandatory to set: {missing}")
        return cls(**merged)
