from .drt1d import DRT  # noqa: F401
from . import qphb  # noqa: F401
