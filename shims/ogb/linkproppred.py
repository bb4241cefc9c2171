"""`from ogb.linkproppred import PygLinkPropPredDataset, Evaluator` (NeighborOverlap_large.py:11)."""
from ocn_amd.evaluate import Evaluator  # noqa: F401


class PygLinkPropPredDataset:
    def __init__(self, *a, **k):
        raise RuntimeError("no network on this box: OGB datasets cannot be downloaded; ogbdataset.loaddataset serves "
                           "a seeded synthetic graph of the named dataset's shape")
