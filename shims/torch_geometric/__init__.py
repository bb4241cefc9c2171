"""Import-resolving stand-in for the slice of torch_geometric the drivers name (NeighborOverlap_large.py:7,12)."""
from . import transforms, utils  # noqa: F401
