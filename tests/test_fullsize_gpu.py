"""Full-size runs of the BASELINE.json configurations on the GPU (`-m gpu`): sizes the oracle cannot finish in
seconds, so the checks are size-independent properties of the domain — the three sweep strategies of the walk
route agree bit for bit, histogram mass equals the per-edge counts, batch-permutation equivariance — plus the
oracle on the first 256 candidates of each batch (as a batch of their own: the normalisation is batch-coupled)."""
from types import SimpleNamespace

import pytest
import torch

from oracle import ocn_oracle as O

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _full_graph(name):
    from ocn_amd.sparse import SparseTensor
    from ocn_amd.synth import dataset_like
    ei, n, shape = dataset_like(name, seed=0)
    adj = SparseTensor.from_edge_index(ei.to(DEV), sparse_sizes=(n, n), trust_data=True).to_symmetric()
    return n, shape, adj


@pytest.fixture(scope="module")
def collab(hiplib):
    from tests.helpers import product_adj2
    n, shape, adj = _full_graph("collab")
    return SimpleNamespace(n=n, adj=adj, adj2=product_adj2(adj))


def test_graph_capture_of_a_65536_edge_step_replays_bitwise(collab):
    """One 65 536-edge cn5 step (source-sorted order, class-major heads, zero-row skipping: every path that only
    runs at B >= 4096) captured as a HIP graph and replayed on fresh candidate ids == the eager call, bit for bit."""
    import ocn_amd.model as M
    from ocn_amd.pipeline import GraphedScorer
    from ocn_amd.synth import sample_edges
    from ocn_amd.utils import adjoverlap
    c = collab
    H, B = 256, 65536
    torch.manual_seed(3)
    h = torch.randn(c.n, H, device=DEV)
    pred = M.predictor_dict["cn5"](H, H, 1, 3, 0.05, 0.4, True).to(DEV).eval()
    r, col, _ = c.adj.coo()
    rc, cc = r.cpu(), col.cpu()
    scorer = GraphedScorer(pred, h, c.adj, c.adj2, B, SimpleNamespace(sum=1.0))
    for seed in (1, 2, 3):
        e = sample_edges(rc, cc, c.n, B, seed=seed).to(DEV)
        got = scorer(e, check=True).clone()
        with torch.no_grad():
            want = pred(h, c.adj, adjoverlap(c.adj, c.adj, e), adjoverlap(c.adj, c.adj2, e), e, None)
        assert torch.equal(got, want)
