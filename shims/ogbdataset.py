"""`from ogbdataset import loaddataset` (NeighborOverlap_large.py:16): same return contract as the reference's
ogbdataset.loaddataset(name, use_valedges_as_input, load=None) (/ogbdataset.py:29-71) on a seeded synthetic graph."""
import os

import torch

from ocn_amd.synth import loaddataset_like


class Data:
    """The attributes the drivers read: x, adj_t, full_adj_t, edge_index, num_nodes, num_features, max_x; .to(device)."""

    def __init__(self, ns):
        self.__dict__.update(vars(ns))
        self.num_features = self.x.shape[-1] if self.x.dim() > 1 else 1

    def to(self, device):
        for k, v in list(self.__dict__.items()):
            if torch.is_tensor(v) or hasattr(v, "to_device"):
                self.__dict__[k] = v.to(device)
        return self


def loaddataset(name: str, use_valedges_as_input: bool, load=None):
    shape = {"Cora": "cora", "Citeseer": "cora", "Pubmed": "cora"}.get(name, name)
    scale = float(os.environ.get("OCN_SYNTH_SCALE", "1.0"))
    data, split_edge = loaddataset_like(shape, use_valedges_as_input, seed=0, scale=scale)
    return Data(data), split_edge
