from .drtmd import (auto_inflight, fit_observations, fit_observations_pfrt, fit_observations_sharded,  # noqa: F401
                    shard_indices)
