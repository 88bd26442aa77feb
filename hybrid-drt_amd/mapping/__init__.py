from .drtmd import (auto_inflight, fit_observations, fit_observations_pfrt, fit_observations_sharded,  # noqa: F401
                    shard_indices)
from .store import DRTMD  # noqa: F401,E402
