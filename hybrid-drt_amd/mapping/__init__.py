from .drtmd import fit_observations, fit_observations_sharded, shard_indices  # noqa: F401
