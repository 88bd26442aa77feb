"""CONTAINER-ONLY SHIM: skimage is imported at module load by hybdrt/filters/_filters.py:3 but never
called on the hot path."""
