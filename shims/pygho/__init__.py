"""Import-only stub: the ppa / citation2 drivers build get_cn1_cn2 from pygho ops themselves
(NeighborOverlap_large_ppa.py:147-173); with ocn_amd use `from utils import get_cn1_cn2` instead."""


class SparseTensor:
    def __init__(self, *a, **k):
        raise NotImplementedError("pygho is not emulated: use ocn_amd.utils.get_cn1_cn2 (shims/utils.py) for the walk-count route")
