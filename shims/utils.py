"""`from utils import ...` of the reference drivers (NeighborOverlap_large.py:14,18,19) -> ocn_amd.utils."""
from ocn_amd.utils import (PermIterator, adjoverlap, block_matrix_multiply, get_cn1_cn2,  # noqa: F401
                           sparse_tensor_multiply)
