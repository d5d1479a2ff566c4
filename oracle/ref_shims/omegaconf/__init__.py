"""Minimal stand-in for the config containers the reference's Constructor uses (see README.md)."""
import dataclasses

MISSING = "???"


def _wrap(v):
    if isinstance(v, DictConfig) or isinstance(v, ListConfig):
        return v
    if isinstance(v, dict):
        return DictConfig(v)
    if isinstance(v, (list, tuple)):
        return ListConfig(v)
    return v


class DictConfig(dict):
    def __init__(self, content=None, **kw):
        super().__init__()
        for k, v in dict(content or {}, **kw).items():
            self[k] = v

    def __setitem__(self, k, v):
        super().__setitem__(k, _wrap(v))

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    def _get_flag(self, name):
        return False


class ListConfig(list):
    def __init__(self, content=()):
        super().__init__(_wrap(v) for v in content)


def _as_dict(c):
    if dataclasses.is_dataclass(c) and not isinstance(c, type):
        return dict(vars(c))
    return c


def _merge_into(dst, src):
    for k, v in _as_dict(src).items():
        v = _as_dict(v)
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            _merge_into(dst[k], v)
        elif isinstance(v, dict):
            dst[k] = DictConfig()
            _merge_into(dst[k], v)
        else:
            dst[k] = v
    return dst


class OmegaConf:
    @staticmethod
    def merge(*configs):
        out = DictConfig()
        for c in configs:
            _merge_into(out, c)
        return out

    @staticmethod
    def create(obj=None):
        return _wrap(obj if obj is not None else {})

    @staticmethod
    def to_container(cfg, resolve=True):
        if isinstance(cfg, dict):
            return {k: OmegaConf.to_container(v) for k, v in cfg.items()}
        if isinstance(cfg, list):
            return [OmegaConf.to_container(v) for v in cfg]
        return cfg

    @staticmethod
    def set_readonly(cfg, value):
        return None

    @staticmethod
    def register_new_resolver(*a, **k):
        return None
