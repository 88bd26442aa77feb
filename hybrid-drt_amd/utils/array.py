"""Host-side grid helpers with the reference's decision rules (hybdrt/utils/array.py:23-45, 142-161).
They only decide *which* kernel variant runs (Toeplitz shortcut or not); no matrix arithmetic happens here."""
import numpy as np


def rel_round(x, precision):
    """Round to `precision` significant digits (hybdrt/utils/array.py:23-45)."""
    x = np.asarray(x, dtype=float)
    scale = np.floor(np.log10(np.abs(x) + 1e-30))
    digits = (precision - scale).astype(int)
    flat = [round(float(v), int(d)) for v, d in zip(x.ravel(), digits.ravel())]
    return np.array(flat).reshape(x.shape)


def is_uniform(x):
    """hybdrt/utils/array.py:142-152."""
    dx = np.diff(x)
    return bool(np.std(dx) / np.mean(dx) <= 0.01)


def is_log_uniform(x):
    """hybdrt/utils/array.py:155-161."""
    return is_uniform(np.log(x))
