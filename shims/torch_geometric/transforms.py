"""`import torch_geometric.transforms as T` (NeighborOverlap_large.py:7): named by the drivers, used only inside the
reference's ogbdataset.py (T.ToSparseTensor), which shims/ogbdataset.py replaces."""


class ToSparseTensor:
    def __init__(self, *a, **k):
        pass

    def __call__(self, data):
        return data


class RandomLinkSplit:
    def __init__(self, *a, **k):
        raise NotImplementedError("datasets are synthetic here: use ogbdataset.loaddataset")
