from .constructor import Constructor, ModuleConfig  # noqa: F401
