from . import basis, mat1d  # noqa: F401
