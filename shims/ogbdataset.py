"""`from ogbdataset import loaddataset` (NeighborOverlap_large.py:16): same return contract as the reference's
ogbdataset.loaddataset(name, use_valedges_as_input, load=None) (/ogbdataset.py:29-71) on a seeded synthetic graph —
served only after an explicit opt-in (OCN_SYNTH=1) and announced on stderr, so that the Hits@K / MRR a driver then prints
cannot be mistaken for results on the real dataset."""
import os
import sys

import torch

from _shimguard import require_synth
from ocn_amd.synth import loaddataset_like

SEED = 0
# shapes ocn_amd.synth knows; Citeseer / Pubmed have no shape of their own there
SHAPES = {"Cora": "cora", "collab": "collab", "ppa": "ppa", "citation2": "citation2", "ddi": "ddi"}


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
    require_synth(f"ogbdataset.loaddataset({name!r})")
    if name not in SHAPES:
        raise ValueError(f"ogbdataset.loaddataset({name!r}): no synthetic shape for this dataset (known: {sorted(SHAPES)}); "
                         "a graph of another dataset's shape under this name would only mislead")
    scale = float(os.environ.get("OCN_SYNTH_SCALE", "1.0"))
    print(f"[ocn shims] SYNTHETIC DATA: '{name}' is a seeded Chung-Lu graph of the {SHAPES[name]} shape (seed {SEED}, scale {scale}); "
          "metrics printed on it are not results on the real dataset", file=sys.stderr, flush=True)
    data, split_edge = loaddataset_like(SHAPES[name], use_valedges_as_input, seed=SEED, scale=scale)
    return Data(data), split_edge
