"""Import-resolving stand-in for the slice of torch_geometric the drivers name (NeighborOverlap_large.py:7,12)."""
from _shimguard import shadowing as _shadowing

_shadowing("torch_geometric")
from . import transforms, utils  # noqa: F401
