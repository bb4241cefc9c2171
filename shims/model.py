"""`from model import ...` of the reference drivers (NeighborOverlap_large.py:8) -> ocn_amd.model."""
from ocn_amd.model import *  # noqa: F401,F403
from ocn_amd.model import (GCN, GCN2, GCN3, DropAdj, DropEdge, GCNConv, PureConv, PureConv2, PureConv3, convdict, convdict2,  # noqa: F401
                           convdict3, predictor_dict)
