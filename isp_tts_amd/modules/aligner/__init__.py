from .mas import b_mas, mas_device  # noqa: F401
