"""Shared builders for the parity tests: the same seeded graph as an oracle SpM and as the product
SparseTensor on the GPU."""
import torch

from oracle import ocn_oracle as O
from ocn_amd.synth import chung_lu_graph, sample_edges


def make_graph(n, avg_deg, max_deg, seed, clique_frac=0.5, isolated=0):
    """Oracle-side symmetric adjacency; the last ``isolated`` node ids keep degree 0."""
    m = n - isolated
    ei = chung_lu_graph(m, avg_deg=avg_deg, max_deg=min(max_deg, m - 1), seed=seed, clique_frac=clique_frac)
    return O.to_symmetric(O.from_edge_index(ei, n))


def to_product(oadj, dev):
    from ocn_amd.sparse import SparseTensor
    return SparseTensor.from_edge_index(torch.stack([oadj.row, oadj.col]).to(dev),
                                        sparse_sizes=(oadj.n_rows, oadj.n_cols))


def product_adj2(adj):
    from ocn_amd.sparse import SparseTensor
    sp = adj.to_torch_sparse_coo_tensor()
    return SparseTensor.from_torch_sparse_coo_tensor(sp @ sp, False)


def batch(oadj, B, seed):
    return sample_edges(oadj.row, oadj.col, oadj.n_rows, B, seed=seed)


def spm_equal(prod, spm):
    """product SparseTensor (device) vs oracle SpM: identical pattern."""
    r, c, _ = prod.coo()
    return (r.cpu().tolist() == spm.row.tolist()) and (c.cpu().tolist() == spm.col.tolist())


def close(a, b, atol=1e-5, rtol=1e-5):
    return torch.allclose(a.detach().cpu(), b.detach().cpu(), atol=atol, rtol=rtol)
