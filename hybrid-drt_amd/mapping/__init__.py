from .drtmd import auto_inflight, fit_observations, fit_observations_sharded, shard_indices  # noqa: F401
