from .drtmd import fit_observations  # noqa: F401
