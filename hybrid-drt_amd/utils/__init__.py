from . import array  # noqa: F401
