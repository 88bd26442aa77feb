"""CONTAINER-ONLY: make /root/reference importable under Python 3.10 (backfill enum.StrEnum used at
hybdrt/dataload/core.py:3) with the shims in this directory first on sys.path."""
import enum
import os
import sys

if not hasattr(enum, "StrEnum"):
    class StrEnum(str, enum.Enum):
        def __str__(self):
            return str(self.value)
    enum.StrEnum = StrEnum

_here = os.path.dirname(os.path.abspath(__file__))
_repo = os.path.dirname(os.path.dirname(_here))
for p in ("/root/reference", _here, _repo):
    if p not in sys.path:
        sys.path.insert(0, p)
